// Match-table consumer stage on the device for gfx950 (MI355X).
//
// Replaces, for a whole scene at once, the per-image / per-pair Python loops of
//   Match2Kpts.__getitem__                  src/coarse_match/utils/merge_kpts.py:36-61
//   agg_groupby_2d(agg="sum")               src/coarse_match/utils/merge_kpts.py:4-17
//   keypoint_worker                         src/coarse_match/coarse_match_worker.py:151-175
//   update_matches(merge=False)             src/coarse_match/coarse_match_worker.py:182-243
//   transform_keypoints                     src/coarse_match/coarse_match_worker.py:250-270
// as called from src/coarse_match/coarse_match.py:203-237: every match endpoint becomes an integer keypoint
// (x, y truncated) of its image; equal keypoints of an image are merged with their confidences summed (float64,
// table order, like np.bincount); an image's keypoints are numbered by descending summed score, ties in (x, y)
// lexicographic order (np.unique order under Python's stable sort); matches are rewritten as keypoint-id pairs.
//
// It is a sort / run-length / segmented-sum problem on 2M entries -- integer, HBM-bound work:
//   key = image << 40 | x << 20 | y  ->  stable radix sort (rocPRIM)  ->  run heads + scan = group ids  ->
//   per-group float64 sums in entry order  ->  stable sort of the groups by score (descending), then by image
//   ->  rank inside the image = keypoint id  ->  scatter ids back to the match rows.
// No host synchronisation: the number of groups stays on the device; the sorts run over the 2M-entry capacity
// with sentinel keys for the unused tail.
#include "common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace {

using namespace dfsfm;

constexpr int COORD_BITS = 20, IMG_SHIFT = 40;
constexpr uint64_t COORD_MASK = (1ull << COORD_BITS) - 1;

__global__ __launch_bounds__(256) void mk_build(const float* __restrict__ rows, const int32_t* __restrict__ img0,
                                                const int32_t* __restrict__ img1, int64_t E, int n_images,
                                                uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                int32_t* __restrict__ err) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int64_t r = e >> 1;
    const int s = (int)(e & 1);
    const int im = s ? img1[r] : img0[r];
    const int x = (int)rows[r * 5 + 2 * s], y = (int)rows[r * 5 + 2 * s + 1];     // astype(int): truncation
    if (im < 0 || im >= n_images || x < 0 || y < 0 || x > (int)COORD_MASK || y > (int)COORD_MASK) *err = 1;
    keys[e] = ((uint64_t)(uint32_t)im << IMG_SHIFT) | ((uint64_t)(x & (int)COORD_MASK) << COORD_BITS) |
              (uint64_t)(y & (int)COORD_MASK);
    vals[e] = (uint32_t)e;
}

__global__ __launch_bounds__(256) void mk_heads(const uint64_t* __restrict__ keys, int64_t E, int32_t* __restrict__ head) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p < E) head[p] = (p == 0 || keys[p] != keys[p - 1]) ? 1 : 0;
}

// One thread per run head: float64 sum of the run's confidences in entry order (np.bincount(weights=...)).
__global__ __launch_bounds__(256) void mk_groups(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                 const int32_t* __restrict__ head, const int32_t* __restrict__ gid,
                                                 const float* __restrict__ rows, int64_t E, double* __restrict__ gsum,
                                                 uint64_t* __restrict__ gkey, uint64_t* __restrict__ score_key,
                                                 uint32_t* __restrict__ gidx, int64_t* __restrict__ n_groups) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= E) return;
    if (p == E - 1) *n_groups = (int64_t)gid[p];            // gid = inclusive scan of the heads: 1-based group number
    if (!head[p]) return;
    double s = 0.0;
    int64_t q = p;
    do {
        s += (double)rows[(int64_t)(vals[q] >> 1) * 5 + 4];
        ++q;
    } while (q < E && !head[q]);
    const int32_t g = gid[p] - 1;
    gsum[g] = s;
    gkey[g] = keys[p];
    score_key[g] = (uint64_t)__double_as_longlong(s) + 1ull;       // s >= 0: the bit pattern orders like the value
    gidx[g] = (uint32_t)g;
}

// Slots >= number of groups of the group arrays: score key 0 sorts last in the descending sort.
__global__ __launch_bounds__(256) void mk_fill_tail(int64_t E, const int64_t* __restrict__ n_groups,
                                                    uint64_t* __restrict__ score_key, uint32_t* __restrict__ gidx) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p < E && p >= *n_groups) {
        score_key[p] = 0ull;
        gidx[p] = 0xFFFFFFFFu;
    }
}

__global__ __launch_bounds__(256) void mk_img_keys(const uint32_t* __restrict__ gidx, const uint64_t* __restrict__ gkey,
                                                   int64_t E, const int64_t* __restrict__ n_groups,
                                                   uint32_t* __restrict__ img_key, int64_t* __restrict__ counts) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= E) return;
    if (p < *n_groups) {
        const uint32_t im = (uint32_t)(gkey[gidx[p]] >> IMG_SHIFT);
        img_key[p] = im;
        atomicAdd(reinterpret_cast<unsigned long long*>(&counts[im + 1]), 1ull);
    } else {
        img_key[p] = 0xFFFFFFFFu;
    }
}

__global__ void mk_offsets(int64_t* __restrict__ offsets, int n_images) {   // in-place inclusive scan of [0, c0, c1, ..]
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int64_t run = 0;
        for (int i = 0; i <= n_images; ++i) {
            run += offsets[i];
            offsets[i] = run;
        }
    }
}

__global__ __launch_bounds__(256) void mk_emit(const uint32_t* __restrict__ gidx, const uint32_t* __restrict__ img_key,
                                               const uint64_t* __restrict__ gkey, const double* __restrict__ gsum,
                                               const int64_t* __restrict__ offsets, int64_t E,
                                               const int64_t* __restrict__ n_groups, float* __restrict__ kpts,
                                               float* __restrict__ scores, int32_t* __restrict__ rank) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= E || p >= *n_groups) return;
    const uint32_t g = gidx[p];
    const uint64_t k = gkey[g];
    kpts[p * 2 + 0] = (float)(int)((k >> COORD_BITS) & COORD_MASK);
    kpts[p * 2 + 1] = (float)(int)(k & COORD_MASK);
    scores[p] = (float)gsum[g];
    rank[g] = (int32_t)(p - offsets[img_key[p]]);
}

__global__ __launch_bounds__(256) void mk_ids(const uint32_t* __restrict__ vals, const int32_t* __restrict__ gid,
                                              const int32_t* __restrict__ rank, int64_t E, int64_t* __restrict__ match_ids) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p < E) match_ids[vals[p]] = (int64_t)rank[gid[p] - 1];
}

struct Plan {
    size_t off_keys[2], off_vals[2], off_head, off_gid, off_gsum, off_gkey, off_skey[2], off_gidx[2], off_ikey[2], off_gidx3,
        off_rank, off_err, off_temp, temp_bytes, total;
};

Plan make_plan(int64_t M) {
    Plan p{};
    const size_t E = (size_t)(2 * M);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    for (int i = 0; i < 2; ++i) p.off_keys[i] = take(E * 8);
    for (int i = 0; i < 2; ++i) p.off_vals[i] = take(E * 4);
    p.off_head = take(E * 4); p.off_gid = take(E * 4); p.off_gsum = take(E * 8); p.off_gkey = take(E * 8);
    for (int i = 0; i < 2; ++i) p.off_skey[i] = take(E * 8);
    for (int i = 0; i < 2; ++i) p.off_gidx[i] = take(E * 4);
    for (int i = 0; i < 2; ++i) p.off_ikey[i] = take(E * 4);
    p.off_gidx3 = take(E * 4); p.off_rank = take(E * 4); p.off_err = take(256);
    size_t t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    uint64_t* k64 = nullptr; uint32_t* v32 = nullptr; int32_t* i32 = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, t1, k64, k64, v32, v32, E, 0, 64, (hipStream_t)0);
    (void)rocprim::radix_sort_pairs_desc(nullptr, t2, k64, k64, v32, v32, E, 0, 64, (hipStream_t)0);
    (void)rocprim::radix_sort_pairs(nullptr, t3, v32, v32, v32, v32, E, 0, 32, (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, t4, i32, i32, E, rocprim::plus<int32_t>(), (hipStream_t)0);
    p.temp_bytes = t1 > t2 ? t1 : t2;
    if (t3 > p.temp_bytes) p.temp_bytes = t3;
    if (t4 > p.temp_bytes) p.temp_bytes = t4;
    p.off_temp = take(p.temp_bytes + 256);
    p.total = off;
    return p;
}

}  // namespace

extern "C" size_t dfsfm_merge_keypoints_workspace(int64_t M) {
    if (M <= 0) return 256;
    return make_plan(M).total;
}

extern "C" int dfsfm_merge_keypoints(const float* rows, const int32_t* img0, const int32_t* img1, int64_t M,
                                     int n_images, float* kpts, float* scores, int64_t* offsets, int64_t* match_ids,
                                     int64_t* n_kpts, int32_t* status, void* workspace, size_t workspace_bytes,
                                     void* stream_) {
    if (M < 0 || n_images <= 0 || n_images >= (1 << 24) || !offsets || !n_kpts || !status) return DFSFM_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (hipMemsetAsync(offsets, 0, (size_t)(n_images + 1) * 8, stream) != hipSuccess ||
        hipMemsetAsync(n_kpts, 0, 8, stream) != hipSuccess || hipMemsetAsync(status, 0, 4, stream) != hipSuccess)
        return DFSFM_E_LAUNCH;
    if (M == 0) return DFSFM_OK;
    if (!rows || !img0 || !img1 || !kpts || !scores || !match_ids || !workspace) return DFSFM_E_BADARG;
    if (2 * M >= (int64_t)0x7FFFFFF0) return DFSFM_E_UNSUPPORTED;
    const Plan p = make_plan(M);
    if (workspace_bytes < p.total) return DFSFM_E_WORKSPACE;
    char* w = static_cast<char*>(workspace);
    const int64_t E = 2 * M;
    const size_t Es = (size_t)E;
    auto u64 = [&](size_t o) { return reinterpret_cast<uint64_t*>(w + o); };
    auto u32 = [&](size_t o) { return reinterpret_cast<uint32_t*>(w + o); };
    auto i32 = [&](size_t o) { return reinterpret_cast<int32_t*>(w + o); };
    void* temp = w + p.off_temp;
    size_t tb = p.temp_bytes;
    const dim3 grid((unsigned)((E + 255) / 256)), blk(256);

    hipLaunchKernelGGL(mk_build, grid, blk, 0, stream, rows, img0, img1, E, n_images, u64(p.off_keys[0]),
                       u32(p.off_vals[0]), status);
    if (rocprim::radix_sort_pairs(temp, tb, u64(p.off_keys[0]), u64(p.off_keys[1]), u32(p.off_vals[0]),
                                  u32(p.off_vals[1]), Es, 0, 64, stream) != hipSuccess)
        return DFSFM_E_LAUNCH;
    hipLaunchKernelGGL(mk_heads, grid, blk, 0, stream, u64(p.off_keys[1]), E, i32(p.off_head));
    tb = p.temp_bytes;
    if (rocprim::inclusive_scan(temp, tb, i32(p.off_head), i32(p.off_gid), Es, rocprim::plus<int32_t>(), stream) !=
        hipSuccess)
        return DFSFM_E_LAUNCH;
    hipLaunchKernelGGL(mk_groups, grid, blk, 0, stream, u64(p.off_keys[1]), u32(p.off_vals[1]), i32(p.off_head),
                       i32(p.off_gid), rows, E, reinterpret_cast<double*>(w + p.off_gsum), u64(p.off_gkey),
                       u64(p.off_skey[0]), u32(p.off_gidx[0]), n_kpts);
    hipLaunchKernelGGL(mk_fill_tail, grid, blk, 0, stream, E, n_kpts, u64(p.off_skey[0]), u32(p.off_gidx[0]));
    tb = p.temp_bytes;
    if (rocprim::radix_sort_pairs_desc(temp, tb, u64(p.off_skey[0]), u64(p.off_skey[1]), u32(p.off_gidx[0]),
                                       u32(p.off_gidx[1]), Es, 0, 64, stream) != hipSuccess)
        return DFSFM_E_LAUNCH;
    hipLaunchKernelGGL(mk_img_keys, grid, blk, 0, stream, u32(p.off_gidx[1]), u64(p.off_gkey), E, n_kpts,
                       u32(p.off_ikey[0]), offsets);
    tb = p.temp_bytes;
    if (rocprim::radix_sort_pairs(temp, tb, u32(p.off_ikey[0]), u32(p.off_ikey[1]), u32(p.off_gidx[1]),
                                  u32(p.off_gidx3), Es, 0, 32, stream) != hipSuccess)
        return DFSFM_E_LAUNCH;
    hipLaunchKernelGGL(mk_offsets, dim3(1), dim3(1), 0, stream, offsets, n_images);
    hipLaunchKernelGGL(mk_emit, grid, blk, 0, stream, u32(p.off_gidx3), u32(p.off_ikey[1]), u64(p.off_gkey),
                       reinterpret_cast<const double*>(w + p.off_gsum), offsets, E, n_kpts, kpts, scores,
                       i32(p.off_rank));
    hipLaunchKernelGGL(mk_ids, grid, blk, 0, stream, u32(p.off_vals[1]), i32(p.off_gid), i32(p.off_rank), E, match_ids);
    return check_launch("dfsfm_merge_keypoints");
}
