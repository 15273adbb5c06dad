// Micro-test (GPU box): which part of a raw-buffer address takes part in the out-of-range check on gfx950?
//   hipcc --offload-arch=gfx950 -O2 tools/micro/buffer_oor.hip -o /tmp/buffer_oor && /tmp/buffer_oor
// Buffer of N bytes followed by N bytes of 0x7f7f7f7f in the same allocation; num_records = N.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k(const char* x, unsigned* out, unsigned n, unsigned voff, unsigned soff)
{
    __shared__ char smem[2048];
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, n, 0x00020000);
    // (a) plain buffer load
    u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff + threadIdx.x * 16, soff, 0);
    // (b) LDS-DMA
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, voff + threadIdx.x * 16, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x * 2] = v.x;
    out[threadIdx.x * 2 + 1] = *(unsigned*)(smem + threadIdx.x * 16);
}
int main()
{
    const unsigned N = 1 << 20;
    char* d; unsigned* o;
    hipMalloc(&d, 2 * N); hipMalloc(&o, 64 * 2 * 4);
    std::vector<unsigned> h(2 * N / 4, 0x11111111u);
    for (size_t i = N / 4; i < h.size(); ++i) h[i] = 0x7f7f7f7fu;
    hipMemcpy(d, h.data(), 2 * N, hipMemcpyHostToDevice);
    struct { unsigned voff, soff; const char* what; } cases[] = {
        {0, 0, "in range"}, {N, 0, "voffset = N"}, {0, N, "soffset = N"}, {N - 512, 0, "voffset straddles the end (lanes 32.. out)"},
        {0, N - 512, "soffset straddles the end"}, {N / 2, N / 2, "voffset + soffset = N"}, {0xfffffc00u, 0x400, "voffset + soffset wraps to 0"}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, N, c.voff, c.soff);
        unsigned r[128];
        hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        printf("%-46s load: lane0 %08x lane63 %08x | lds-dma: lane0 %08x lane63 %08x\n", c.what, r[0], r[126], r[1], r[127]);
    }
    return 0;
}
