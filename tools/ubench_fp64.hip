// Micro-benchmark: FP64 VALU issue/latency for ONE wave per SIMD on gfx950 (the regime the RK4 fan kernel runs in).
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench_fp64 tools/ubench_fp64.hip ; run: ./ubench_fp64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

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

template <int CHAINS, int OP>
__global__ void k_op(double* out, int iters, double a){
    double x[CHAINS];
    #pragma unroll
    for(int c = 0; c < CHAINS; c++) x[c] = 1.5 + threadIdx.x * 1e-3 + c;
    for(int i = 0; i < iters; i++){
        #pragma unroll
        for(int r = 0; r < 4; r++){
            #pragma unroll
            for(int c = 0; c < CHAINS; c++){
                if(OP == 0) x[c] = 1.0 / x[c] + a;                 // IEEE divide
                if(OP == 1) x[c] = sqrt(x[c]) + a;                 // IEEE sqrt
                if(OP == 2) x[c] = __builtin_amdgcn_rcp(x[c]) + a; // raw v_rcp_f64
                if(OP == 3) x[c] = __builtin_amdgcn_rsq(x[c]) + a; // raw v_rsq_f64
                if(OP == 4) x[c] = exp(-x[c]) + a;
                if(OP == 5) { double s_ = sin(x[c]); x[c] = s_ + a; }
            }
        }
    }
    double s = 0;
    #pragma unroll
    for(int c = 0; c < CHAINS; c++) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float timeit(F f){
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main(){
    double* out; hipMalloc(&out, sizeof(double) * 1024 * 1024);
    const int iters = 20000;
    // blocks of 64 threads: 256 blocks -> 1 wave per CU ; 1024 -> 1 wave/SIMD ; 2048 -> 2 waves/SIMD
    int grids[] = {256, 1024, 2048, 4096};
    printf("v_fma_f64: ns per wave-instruction (lower = faster issue); clock ~2.4 GHz => cycles = ns*2.4\n");
    for(int g : grids){
        float m1 = timeit([&]{ hipLaunchKernelGGL(k_fma<1>, dim3(g), dim3(64), 0, 0, out, iters, 0.999, 0.001); });
        float m2 = timeit([&]{ hipLaunchKernelGGL(k_fma<2>, dim3(g), dim3(64), 0, 0, out, iters, 0.999, 0.001); });
        float m4 = timeit([&]{ hipLaunchKernelGGL(k_fma<4>, dim3(g), dim3(64), 0, 0, out, iters, 0.999, 0.001); });
        float m8 = timeit([&]{ hipLaunchKernelGGL(k_fma<8>, dim3(g), dim3(64), 0, 0, out, iters, 0.999, 0.001); });
        double n = (double)iters * 16;
        printf("grid %5d waves: chains1 %.2f ns  chains2 %.2f  chains4 %.2f  chains8 %.2f (per instr)\n", g,
               m1 * 1e6 / n, m2 * 1e6 / (2 * n), m4 * 1e6 / (4 * n), m8 * 1e6 / (8 * n));
    }
    const char* names[] = {"1.0/x (IEEE)", "sqrt (IEEE)", "v_rcp_f64", "v_rsq_f64", "exp", "sin"};
    const int it2 = 5000;
    for(int g : {1024, 2048}){
        printf("grid %d waves, ns per op: ", g);
        float t;
        t = timeit([&]{ hipLaunchKernelGGL((k_op<1,0>), dim3(g), dim3(64), 0, 0, out, it2, 0.5); }); printf("%s dep %.1f ", names[0], t * 1e6 / (it2 * 4.0));
        t = timeit([&]{ hipLaunchKernelGGL((k_op<4,0>), dim3(g), dim3(64), 0, 0, out, it2, 0.5); }); printf("indep4 %.1f | ", t * 1e6 / (it2 * 16.0));
        t = timeit([&]{ hipLaunchKernelGGL((k_op<1,1>), dim3(g), dim3(64), 0, 0, out, it2, 0.5); }); printf("%s dep %.1f ", names[1], t * 1e6 / (it2 * 4.0));
        t = timeit([&]{ hipLaunchKernelGGL((k_op<4,1>), dim3(g), dim3(64), 0, 0, out, it2, 0.5); }); printf("indep4 %.1f | ", t * 1e6 / (it2 * 16.0));
        t = timeit([&]{ hipLaunchKernelGGL((k_op<1,2>), dim3(g), dim3(64), 0, 0, out, it2, 0.5); }); printf("%s dep %.1f ", names[2], t * 1e6 / (it2 * 4.0));
        t = timeit([&]{ hipLaunchKernelGGL((k_op<4,2>), dim3(g), dim3(64), 0, 0, out, it2, 0.5); }); printf("indep4 %.1f | ", t * 1e6 / (it2 * 16.0));
        t = timeit([&]{ hipLaunchKernelGGL((k_op<1,3>), dim3(g), dim3(64), 0, 0, out, it2, 0.5); }); printf("%s dep %.1f ", names[3], t * 1e6 / (it2 * 4.0));
        t = timeit([&]{ hipLaunchKernelGGL((k_op<1,4>), dim3(g), dim3(64), 0, 0, out, it2, 0.5); }); printf("| %s dep %.1f ", names[4], t * 1e6 / (it2 * 4.0));
        t = timeit([&]{ hipLaunchKernelGGL((k_op<1,5>), dim3(g), dim3(64), 0, 0, out, it2, 0.5); }); printf("| %s dep %.1f\n", names[5], t * 1e6 / (it2 * 4.0));
    }
    return 0;
}
