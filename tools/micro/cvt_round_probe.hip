// Probe: rounding of v_cvt_pk_u8_f32 and v_cvt_u32_f32 on gfx950 (is the float -> u8 pack a truncation?  does it follow MODE.fp_round?).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int n)
{
    const int i = threadIdx.x;
    if (i >= n) return;
    unsigned r = 0, r2 = 0;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(r) : "v"(in[i]));
    // the same with MODE.fp_round (f32) = 3, round toward zero
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\tv_cvt_pk_u8_f32 %0, %1, 0, %0\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0" : "+v"(r2) : "v"(in[i]));
    out[2 * i] = r | (r2 << 8);
    out[2 * i + 1] = (unsigned)in[i];
}
int main()
{
    const float h[] = {0.f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 2.49f, 2.51f, 3.5f, 254.5f, 255.4f, 255.5f, 255.9f, 256.7f, 300.f, -0.3f, -1.f, 127.99999f};
    const int n = sizeof(h) / sizeof(h[0]);
    float* d; unsigned* o; unsigned ho[2 * n];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("%12.6f -> cvt_pk_u8 %3u   (round mode 3: %3u)   cvt_u32 %3u\n", h[i], ho[2 * i] & 0xff, (ho[2 * i] >> 8) & 0xff, ho[2 * i + 1]);
    return 0;
}
