// Does MODE.FP16_OVFL (bit 23 of HW_REG_MODE) make v_cvt_pk_f16_f32 / v_cvt_f16_f32 saturate at +-65504 instead of overflowing to inf on gfx950?
//   hipcc --offload-arch=gfx950 -O3 fp16_ovfl.hip -o fp16_ovfl && ./fp16_ovfl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, float* out, int set) {
    if (set) __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);
    const float a = in[threadIdx.x * 2], b = in[threadIdx.x * 2 + 1];
    half2_t h = {(_Float16)a, (_Float16)b};
    asm volatile("" : "+v"(h));
    out[threadIdx.x * 2] = (float)h[0];
    out[threadIdx.x * 2 + 1] = (float)h[1];
}
int main() {
    const float hin[8] = {1.0f, 65504.f, 65519.f, 65520.f, 1e6f, -1e6f, -70000.f, __builtin_inff()};
    float *din, *dout, hout[8];
    hipMalloc(&din, 32); hipMalloc(&dout, 32);
    hipMemcpy(din, hin, 32, hipMemcpyHostToDevice);
    for (int set = 0; set < 2; ++set) {
        hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, din, dout, set);
        hipMemcpy(hout, dout, 32, hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d:", set);
        for (int i = 0; i < 8; ++i) printf(" %g->%g", hin[i], hout[i]);
        printf("\n");
    }
    return 0;
}
