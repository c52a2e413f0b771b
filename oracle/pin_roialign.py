"""TEST INFRASTRUCTURE ONLY -- recipe that PINS the RoIAlign stage once its source is reachable.

The reference crops its refinement patches with ``roi_align.RoIAlign`` from the git submodule
``third_party/RoIAlign.pytorch`` (https://github.com/longcw/RoIAlign.pytorch, /root/reference/.gitmodules); the submodule
directory is empty in the reference snapshot and this environment has no network, so ``oracle/restate.py::roi_align_crop``
restates the published algorithm (TensorFlow ``crop_and_resize``) and the stage is "parity unpinned" (DESIGN.md section 4).

This script is the committed way out.  It is OPT-IN: nothing runs it implicitly (``__graft_entry__.build()`` does not, the
tests only ``load()`` a library that already exists), because it compiles and later executes third-party C.
``python -m oracle.pin_roialign [--clone --commit <sha>] [--sha256 <hex>]``

1. looks for the upstream CPU kernel ``roi_align/src/crop_and_resize.c`` under $ROIALIGN_SRC (a checkout directory, or a
   ``.tar.gz`` / ``.tgz`` / ``.zip`` archive of one -- e.g. GitHub's archive of the pinned commit -- which is unpacked into
   ``oracle/_ref/RoIAlign.pytorch.src``; give its hash with --sha256-archive / $ROIALIGN_ARCHIVE_SHA256), then under the submodule
   directory, then -- only with ``--clone --commit <40-hex sha>`` -- fetches exactly that commit into
   ``oracle/_ref/RoIAlign.pytorch`` (no unpinned HEAD is ever cloned); with ``--sha256`` (or $ROIALIGN_SHA256) the file must
   hash to that value before anything is compiled, and the hash of whatever was compiled is printed and stored next to the
   library (``libcrop_and_resize.sha256``) so a reviewer can pin it afterwards;
2. if found, takes the ONE self-contained function of that file, ``CropAndResizePerBox`` (plain C on float / int pointers; the
   rest of the file is TH tensor glue of a 2018 PyTorch), writes it -- from where it lies, at build time, never into the git
   history -- to ``oracle/_ref/crop_and_resize_core.c`` and builds ``oracle/_ref/libcrop_and_resize.so`` with gcc;
3. ``tests/test_roialign_pin.py`` then compares ``restate.roi_align_crop`` with that library bit for bit (skipped while the
   library does not exist).

Exit code 0: library built.  2: source not reachable (the state of this round; the message says what was tried).  3: hash mismatch.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
URL = "https://github.com/longcw/RoIAlign.pytorch.git"
REL = os.path.join("roi_align", "src", "crop_and_resize.c")
LIB = os.path.join(REF_DIR, "libcrop_and_resize.so")


def unpack_archive(path, expect_sha256=None):
    """A source archive supplied through $ROIALIGN_SRC -> directory that contains roi_align/src/crop_and_resize.c (or None).
    Only regular members below the archive's own top directory are extracted (no absolute paths, no '..')."""
    import hashlib
    import tarfile
    import zipfile
    with open(path, "rb") as fh:
        digest = hashlib.sha256(fh.read()).hexdigest()
    if expect_sha256 and digest != expect_sha256.lower():
        raise RuntimeError(f"archive {path}: sha256 {digest} != expected {expect_sha256}")
    dst = os.path.join(REF_DIR, "RoIAlign.pytorch.src")
    os.makedirs(dst, exist_ok=True)

    def safe(name):
        return not (name.startswith("/") or ".." in name.split("/"))
    if zipfile.is_zipfile(path):
        with zipfile.ZipFile(path) as z:
            for n in z.namelist():
                if safe(n) and not n.endswith("/"):
                    z.extract(n, dst)
    else:
        with tarfile.open(path) as t:
            for m in t.getmembers():
                if m.isfile() and safe(m.name):
                    t.extract(m, dst)
    for root, _, files in os.walk(dst):
        if root.endswith(os.path.join("roi_align", "src")) and "crop_and_resize.c" in files:
            return os.path.dirname(os.path.dirname(root))
    return None


def find_source(allow_clone=False, commit=None):
    tried = []
    env = os.environ.get("ROIALIGN_SRC")
    if env and os.path.isfile(env):                       # an archive of the pinned commit
        tried.append(env + " (archive)")
        env = unpack_archive(env, os.environ.get("ROIALIGN_ARCHIVE_SHA256"))
    roots = [env, os.path.join(os.environ.get("DFSFM_REFERENCE_ROOT", "/root/reference"),
                                                            "third_party", "RoIAlign.pytorch"),
             os.path.join(REF_DIR, "RoIAlign.pytorch")]
    for root in roots:
        if not root:
            continue
        path = os.path.join(root, REL)
        tried.append(path)
        if os.path.isfile(path):
            return path, tried
    if not allow_clone:
        return None, tried
    if not (commit and re.fullmatch(r"[0-9a-f]{40}", commit)):
        tried.append("clone refused: --clone needs --commit <40-hex sha> (an unpinned HEAD is never fetched)")
        return None, tried
    os.makedirs(REF_DIR, exist_ok=True)
    dst = os.path.join(REF_DIR, "RoIAlign.pytorch")
    try:
        subprocess.run(["git", "init", "-q", dst], check=True, timeout=20, capture_output=True)
        subprocess.run(["git", "-C", dst, "fetch", "--depth", "1", URL, commit], check=True, timeout=30, capture_output=True)
        subprocess.run(["git", "-C", dst, "checkout", "-q", "FETCH_HEAD"], check=True, timeout=20, capture_output=True)
    except Exception as e:                                      # no network here: say so, do not guess
        tried.append(f"git fetch {URL} {commit} -> {type(e).__name__}")
        return None, tried
    path = os.path.join(dst, REL)
    tried.append(path)
    return (path if os.path.isfile(path) else None), tried


def extract_core(text):
    """The text of ``void CropAndResizePerBox(...) { ... }`` (brace matching; the function has no dependencies but math.h)."""
    m = re.search(r"void\s+CropAndResizePerBox\s*\(", text)
    if not m:
        raise RuntimeError("CropAndResizePerBox not found: upstream layout changed, adapt this recipe")
    i = text.index("{", m.end())
    depth, j = 0, i
    while True:
        depth += {"{": 1, "}": -1}.get(text[j], 0)
        j += 1
        if depth == 0:
            break
    return text[m.start():j]


def main(argv=None):
    import argparse
    import hashlib
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--clone", action="store_true", help="fetch the pinned commit when no local source exists")
    ap.add_argument("--commit", default=None, help="40-hex commit of longcw/RoIAlign.pytorch to fetch with --clone")
    ap.add_argument("--sha256", default=os.environ.get("ROIALIGN_SHA256"), help="required hash of crop_and_resize.c")
    a = ap.parse_args(argv)
    src, tried = find_source(allow_clone=a.clone, commit=a.commit)
    if src is None:
        print("RoIAlign.pytorch source not reachable; a12 stays parity-unpinned.  Tried:")
        for t in tried:
            print("  ", t)
        return 2
    with open(src, "rb") as fh:
        raw = fh.read()
    digest = hashlib.sha256(raw).hexdigest()
    if a.sha256 and digest != a.sha256.lower():
        print(f"refusing to compile {src}: sha256 {digest} != expected {a.sha256}")
        return 3
    core = extract_core(raw.decode("utf-8", "replace"))
    os.makedirs(REF_DIR, exist_ok=True)
    cfile = os.path.join(REF_DIR, "crop_and_resize_core.c")
    with open(cfile, "w") as fh:
        fh.write("/* generated by oracle/pin_roialign.py from " + src + " -- not part of the repository */\n#include <math.h>\n"
                 "#include <stdio.h>\n" + core + "\n")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-ffp-contract=off", cfile, "-o", LIB, "-lm"])
    with open(os.path.join(REF_DIR, "libcrop_and_resize.sha256"), "w") as fh:
        fh.write(f"{digest}  {src}\n")
    print("built", LIB, "from", src, "sha256", digest, "(verified)" if a.sha256 else "(UNVERIFIED: pass --sha256 to pin it)")
    return 0


def load():
    """ctypes handle of the pinned kernel, or None.  Signature (upstream): CropAndResizePerBox(image, batch, depth, H, W, boxes
    [y1,x1,y2,x2 normalised], box_index, start_box, limit_box, crops, crop_h, crop_w, extrapolation_value)."""
    if not os.path.isfile(LIB):
        return None
    import ctypes
    lib = ctypes.CDLL(LIB)
    f = lib.CropAndResizePerBox
    f.restype = None
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                  ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float]
    return f


if __name__ == "__main__":
    sys.exit(main())
