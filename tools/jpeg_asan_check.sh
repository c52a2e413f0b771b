#!/bin/bash
# One-off hardening check of the device JPEG decoder's thread functions: the CPU lane model (tests/jpeg_emul.cpp, the same
# csrc/jpeg_core.h the kernels compile) built with AddressSanitizer and run on the damaged-header / corrupt-scan / every-layout /
# large-frame tests.  numpy's buffers (workspace, output, tables) come from malloc, so a read or write outside them is reported.
# Last run: 4 passed, no report (r05).
cd "$(dirname "$0")/.." || exit 1
keep=$(mktemp); cp tests/_build/libjpeg_emul.so "$keep" 2>/dev/null
g++ -O1 -g -fsanitize=address -fno-omit-frame-pointer -std=c++17 -fPIC -shared -Iinclude -Idetectorfreesfm_amd/csrc tests/jpeg_emul.cpp -o tests/_build/libjpeg_emul.so
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python -m pytest tests/test_jpeg_cpu.py -x -q -k "damaged or corrupt or every_supported or large_frame"
rc=$?
[ -s "$keep" ] && cp "$keep" tests/_build/libjpeg_emul.so; touch tests/_build/libjpeg_emul.so; rm -f "$keep"
exit $rc
