// Pieces shared by the fused encoder-layer kernels (encoder_fused.hip: d_model 128; encoder256.hip: d_model 256): the
// counted VM wait and the weight-slab ring -- a cyclic stream of 16-KB slabs of ready-made MFMA A fragments, DMA'd
// global -> LDS (buffer_load ... lds) by the four waves of a workgroup, one s_barrier per slab.
#pragma once
#include "common.h"

namespace dfsfm_enc {

typedef __attribute__((address_space(3))) void lds_void;

constexpr int SLAB = 16384;              // bytes of one weight slab: 16 A fragments of 1 KB
constexpr int NSTG = 4;                  // ring depth (slabs)
constexpr int RING = NSTG * SLAB;

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---------------------------------------------------------------------------------------------------------------------
// weight-slab ring shared by both kernels: slab g of the cyclic stream lives in stage g % NSTG
// ---------------------------------------------------------------------------------------------------------------------
struct SlabRing {
    char* ring;
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned lane_off;       // wave * 4096 + lane * 16
    int wave, nslab;
    unsigned next;           // next slab index to consume (monotonic)

    __device__ __forceinline__ void issue(unsigned g) const {
        const unsigned src = (g % (unsigned)nslab) * SLAB + lane_off;
        char* dst = ring + (g % NSTG) * SLAB + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(dst + i * 1024), 16, src + i * 1024, 0, 0, 0);
    }
    __device__ __forceinline__ void issue_piece(unsigned g, int i) const {
        const unsigned src = (g % (unsigned)nslab) * SLAB + lane_off;
        char* dst = ring + (g % NSTG) * SLAB + wave * 4096;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(dst + i * 1024), 16, src + i * 1024, 0, 0, 0);
    }
    // acquire without the refill: the caller spreads the four pieces of slab next + NSTG - 1 between its MFMA groups
    // (issue_piece) and then calls advance()
    __device__ __forceinline__ const char* acquire_wait() {
        wait_vmcnt<(NSTG - 2) * 4>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        return ring + (next % NSTG) * SLAB;
    }
    __device__ __forceinline__ void advance() { ++next; }
    __device__ __forceinline__ void prologue() {
#pragma unroll
        for (int g = 0; g < NSTG - 1; ++g) issue(g);
        next = 0;
    }
    // make slab `next` readable by every wave and refill the stage the previous slab occupied; returns its LDS address.
    // Loads complete in issue order, so "at most (NSTG-2)*4 outstanding" means this wave's pieces of the slab have landed
    // (anything issued in between -- other loads, or stores, which may complete out of order with respect to loads but
    // only ever add to the count -- makes the wait more conservative, never less).
    __device__ __forceinline__ const char* acquire() {
        wait_vmcnt<(NSTG - 2) * 4>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue(next + NSTG - 1);
        const char* p = ring + (next % NSTG) * SLAB;
        ++next;
        return p;
    }
};

// chunk partials [N][nchunks][8][32*32 + 32] (K1's layout: KV[d][v], then Ksum[d]) -> the apply image of encoder256.hip
// (defined there; also used by dfsfm_encoder256_state_f32 in linear_attention.hip)
void enc256_launch_image(const float* part, char* img, int N, int nchunks, hipStream_t stream);

}  // namespace dfsfm_enc
