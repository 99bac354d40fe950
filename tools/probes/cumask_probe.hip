// Does a stream created with hipExtStreamCreateWithCUMask confine its kernels to the CUs of the mask on this box, and which CUs are they?
// Every workgroup records (XCC_ID, HW_ID) of its first wave; the host counts distinct (xcc, se, sh, cu).   hipcc --offload-arch=gfx950 cumask_probe.hip -o cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void k_where(unsigned* out, int spin){
    if(threadIdx.x == 0){
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_ID, 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // XCC_ID bits 3:0
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    }
    for(int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(64);                       // (hold the CU a while so that the grid spreads)
}
static void run(hipStream_t s, const char* what){
    const int n = 4096;
    unsigned* d; hipMalloc(&d, sizeof(unsigned) * 2 * n);
    hipLaunchKernelGGL(k_where, dim3(n), dim3(256), 64 * 1024, s, d, 200);
    hipError_t e = hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * n); hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * n, hipMemcpyDeviceToHost);
    std::set<unsigned> cus; std::set<unsigned> xccs;
    for(int i = 0; i < n; i++){
        const unsigned hw = h[2 * i], x = h[2 * i + 1] & 15;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cus.insert((x << 12) | (se << 8) | (sh << 4) | cu); xccs.insert(x);
    }
    printf("%s: %s, distinct CUs %zu on %zu XCCs\n", what, hipGetErrorString(e), cus.size(), xccs.size());
    hipFree(d);
}
int main(){
    hipStream_t s0; hipStreamCreate(&s0); run(s0, "plain stream");
    for(int bits : {32, 64, 88, 128, 168, 256}){
        uint32_t mask[8] = {0};
        for(int i = 0; i < bits; i++) mask[i / 32] |= 1u << (i % 32);
        hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
        char what[64]; snprintf(what, sizeof what, "mask of the first %d bits (create: %s)", bits, hipGetErrorString(e));
        if(e == hipSuccess) run(s, what); else printf("%s\n", what);
    }
    // the complement of the first 88 bits
    { uint32_t mask[8]; for(int w = 0; w < 8; w++) mask[w] = 0xffffffffu; for(int i = 0; i < 88; i++) mask[i / 32] &= ~(1u << (i % 32));
      hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask); if(e == hipSuccess) run(s, "complement of the first 88 bits"); else printf("complement: %s\n", hipGetErrorString(e)); }
    return 0;
}
