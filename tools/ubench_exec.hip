// Micro-benchmark: does a wave with a partly empty EXEC mask issue FP64 VALU faster on gfx950?  (16 / 32 / 64 active lanes, 1 wave per SIMD)
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench_exec tools/ubench_exec.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CHAINS>
__global__ void k_fma(double* out, int iters, double a, double b){
    double x[CHAINS];
    #pragma unroll
    for(int c = 0; c < CHAINS; c++) x[c] = threadIdx.x * 1e-3 + c;
    for(int i = 0; i < iters; i++){
        #pragma unroll
        for(int r = 0; r < 16; r++){
            #pragma unroll
            for(int c = 0; c < CHAINS; c++) x[c] = __builtin_fma(x[c], a, b);
        }
    }
    double s = 0;
    #pragma unroll
    for(int c = 0; c < CHAINS; c++) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> float timeit(F f){
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main(){
    double* out; hipMalloc(&out, sizeof(double) * 1024 * 64);
    const int iters = 20000; const double n = (double)iters * 16;
    for(int lanes : {64, 32, 16, 8}){
        float m1 = timeit([&]{ hipLaunchKernelGGL(k_fma<1>, dim3(1024), dim3(lanes), 0, 0, out, iters, 0.999, 0.001); });
        float m4 = timeit([&]{ hipLaunchKernelGGL(k_fma<4>, dim3(1024), dim3(lanes), 0, 0, out, iters, 0.999, 0.001); });
        printf("%2d active lanes per wave, 1024 waves: dependent chain %.2f ns / instr, 4 chains %.2f ns / instr\n", lanes, m1 * 1e6 / n, m4 * 1e6 / (4 * n));
    }
    return 0;
}
