// Probe: rounding of v_cvt_pk_u8_f32 and v_cvt_u32_f32 on gfx950 (is the float -> u8 pack a truncation?).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int n)
{
    const int i = threadIdx.x;
    if (i >= n) return;
    unsigned r = 0;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(r) : "v"(in[i]));
    out[2 * i] = r;
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
    for (int i = 0; i < n; i++) printf("%12.6f -> cvt_pk_u8 %3u   cvt_u32 %3u\n", h[i], ho[2 * i] & 0xff, ho[2 * i + 1]);
    return 0;
}
