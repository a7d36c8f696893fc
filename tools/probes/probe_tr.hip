// probe: semantics of ds_read_b64_tr_b16 on gfx950 -- every lane supplies an 8-byte-aligned LDS address; what does it get back?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const int *addr, uint16_t *out)
{
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;     // element value = its index
    __syncthreads();
    const uint32_t a = (uint32_t)(uintptr_t)(lds) + addr[threadIdx.x] * 2;  // addr in elements
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = v[0] & 0xffff; out[threadIdx.x * 4 + 1] = v[0] >> 16;
    out[threadIdx.x * 4 + 2] = v[1] & 0xffff; out[threadIdx.x * 4 + 3] = v[1] >> 16;
}
int main()
{
    int h_addr[64]; uint16_t h_out[256];
    int *d_addr; uint16_t *d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int variant = 0; variant < 3; ++variant) {
        for (int l = 0; l < 64; ++l) {
            if (variant == 0) h_addr[l] = 4 * l;                                   // consecutive 8-byte pieces
            else if (variant == 1) h_addr[l] = ((l >> 4) * 4 + ((l & 15) >> 2)) * 100 + 4 * (l & 3);   // rows 100 elements apart: lane -> row 4g + i/4, cols 4 (i%4)
            else h_addr[l] = (l & 15) * 100 + (l >> 4) * 4;                        // every lane its own row
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("variant %d\n", variant);
        for (int l = 0; l < 64; ++l) printf("lane %2d addr %4d -> %4d %4d %4d %4d\n", l, h_addr[l], h_out[4 * l], h_out[4 * l + 1], h_out[4 * l + 2], h_out[4 * l + 3]);
    }
    return 0;
}
