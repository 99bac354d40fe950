// Micro-benchmark for the grid kernels: how fast can a CU gather per-lane records?  Every lane reads whole 320-byte records (20 x 16 B,
// the (field, kz, node) record of geoac_rngdep.h) at its own pseudo-random record index from a table of a given size, through
//   (a) global_load_dwordx4 (divergent: one record per lane), (b) the same with all lanes of a wave on the SAME record,
//   (c) ds_read_b128 from an LDS copy (per-lane records), and reports cycles per wave-instruction and bytes per clock per CU
// at one and two waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 -o ubench_gather tools/ubench_gather.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REC 40            // doubles per record

template <bool UNIFORM>
__global__ void __launch_bounds__(256) k_gather(const double* __restrict__ tab, unsigned nrec, int iters, double* out){
    unsigned idx = (UNIFORM ? (blockIdx.x * 4 + threadIdx.x / 64) : (blockIdx.x * blockDim.x + threadIdx.x)) * 2654435761u;
    double acc = 0.0;
    for(int i = 0; i < iters; i++){
        idx = idx * 1664525u + 1013904223u;
        const double2* r = (const double2*)(tab + (size_t)(idx % nrec) * REC);
        double2 v[20];
        #pragma unroll
        for(int q = 0; q < 20; q++) v[q] = r[q];
        #pragma unroll
        for(int q = 0; q < 20; q++) acc += v[q].x * v[q].y;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// cooperative pattern: groups of G lanes read G x 16 contiguous bytes of ONE owner's record per instruction (the owner rotates over the
// group), so an instruction touches 64 / G distinct 16 G-byte pieces instead of 64 separate lines
template <int G>
__global__ void __launch_bounds__(256) k_coop(const double* __restrict__ tab, unsigned nrec, int iters, double* out){
    unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
    const int lane = threadIdx.x & 63, r = lane % G, base = lane - r;
    double acc = 0.0;
    constexpr int PER = 20 / G > 0 ? 20 / G : 1;                 // pieces of 16 G bytes per record (320 B records; G = 8, 16: the tail piece wraps)
    for(int i = 0; i < iters; i++){
        idx = idx * 1664525u + 1013904223u;
        const unsigned mine = idx % nrec;
        double2 v[20];
        #pragma unroll
        for(int q = 0; q < 20; q++){
            const int owner = base + (q / PER) % G, piece = q % PER;
            const unsigned rec = __shfl(mine, owner);
            v[q] = ((const double2*)(tab + (size_t)rec * REC))[(piece * G + r) % 20];
        }
        #pragma unroll
        for(int q = 0; q < 20; q++) acc += v[q].x * v[q].y;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ void __launch_bounds__(256) k_lds(const double* __restrict__ tab, unsigned nrec, int iters, double* out){
    extern __shared__ double lds[];
    const unsigned nl = 160 * 1024 / 8 / REC / 2;                           // records in (half of) the LDS: two blocks per CU at two waves per SIMD
    for(unsigned q = threadIdx.x; q < nl * REC; q += blockDim.x) lds[q] = tab[q];
    __syncthreads();
    unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
    double acc = 0.0;
    for(int i = 0; i < iters; i++){
        idx = idx * 1664525u + 1013904223u;
        const double2* r = (const double2*)(lds + (size_t)(idx % nl) * REC);
        double2 v[20];
        #pragma unroll
        for(int q = 0; q < 20; q++) v[q] = r[q];
        #pragma unroll
        for(int q = 0; q < 20; q++) acc += v[q].x * v[q].y;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename F> float timeit(F f){
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main(){
    const double ghz = 2.4;
    double* out; hipMalloc(&out, sizeof(double) * 512 * 256);
    const int iters = 2000;
    for(double mb : {2.0, 9.6, 38.0, 150.0}){
        unsigned nrec = (unsigned)(mb * 1e6 / (REC * 8));
        double* tab; hipMalloc(&tab, (size_t)nrec * REC * 8);
        std::vector<double> h((size_t)nrec * REC, 1.0);
        hipMemcpy(tab, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        for(int wps : {1, 2}){
            const int blocks = 256 * wps;                                   // 256-thread blocks: one wave per SIMD per block
            const double winstr = (double)iters * 20;                      // wave-instructions per wave
            float a = timeit([&]{ hipLaunchKernelGGL(k_gather<false>, dim3(blocks), dim3(256), 0, 0, tab, nrec, iters, out); });
            float b = timeit([&]{ hipLaunchKernelGGL(k_gather<true>, dim3(blocks), dim3(256), 0, 0, tab, nrec, iters, out); });
            // per CU: 4 * wps waves, each winstr instructions of 1 KiB
            double cyc_a = a * 1e6 * ghz / (winstr * 4 * wps), cyc_b = b * 1e6 * ghz / (winstr * 4 * wps);
            float c2 = timeit([&]{ hipLaunchKernelGGL(k_coop<2>, dim3(blocks), dim3(256), 0, 0, tab, nrec, iters, out); });
            float c4 = timeit([&]{ hipLaunchKernelGGL(k_coop<4>, dim3(blocks), dim3(256), 0, 0, tab, nrec, iters, out); });
            float c8 = timeit([&]{ hipLaunchKernelGGL(k_coop<8>, dim3(blocks), dim3(256), 0, 0, tab, nrec, iters, out); });
            float c16 = timeit([&]{ hipLaunchKernelGGL(k_coop<16>, dim3(blocks), dim3(256), 0, 0, tab, nrec, iters, out); });
            const double k = 1e6 * ghz / (winstr * 4 * wps);
            printf("table %6.1f MB, %d wave(s)/SIMD: cooperative groups of 2 / 4 / 8 / 16 lanes: %.1f / %.1f / %.1f / %.1f clk per wave-instr per CU\n", mb, wps, c2 * k, c4 * k, c8 * k, c16 * k);
            printf("table %6.1f MB, %d wave(s)/SIMD: per-lane records %.1f clk per wave-instr per CU (%.0f B/clk/CU); wave-uniform record %.1f clk (%.0f B/clk/CU)\n",
                   mb, wps, cyc_a, 1024.0 / cyc_a, cyc_b, 1024.0 / cyc_b);
        }
        hipFree(tab);
    }
    {
        unsigned nrec = 4096; double* tab; hipMalloc(&tab, (size_t)nrec * REC * 8); hipMemset(tab, 0, (size_t)nrec * REC * 8);
        hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        for(int wps : {1, 2}){
            const int blocks = 256 * wps; const double winstr = (double)iters * 20;
            float a = timeit([&]{ hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 80 * 1024, 0, tab, nrec, iters, out); });
            double cyc = a * 1e6 * ghz / (winstr * 4 * wps);
            printf("LDS copy, %d wave(s)/SIMD: per-lane records %.1f clk per wave-instr per CU (%.0f B/clk/CU)\n", wps, cyc, 1024.0 / cyc);
        }
    }
    return 0;
}
