// accuracy of v_rcp_f64 / v_rsq_f64 and Newton-refined forms vs IEEE results (max relative error over a sweep)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(double* out, int n){
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) return;
    double x = 0.001 + 7000.0 * (double)i / n + 1e-7 * i;
    double r_ex = 1.0 / x, s_ex = 1.0 / sqrt(x);
    double r0 = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r0, 1.0); double r1 = __builtin_fma(r0, e, r0);
    e = __builtin_fma(-x, r1, 1.0); double r2 = __builtin_fma(r1, e, r1);
    double y0 = __builtin_amdgcn_rsq(x);
    double g = x * y0; double h = 0.5 * y0; double rr = __builtin_fma(-h, g, 0.5);   // Goldschmidt-style
    double y1 = __builtin_fma(y0, rr, y0);          // y0 * (1 + (0.5 - 0.5 x y0^2)) 
    g = x * y1; h = 0.5 * y1; rr = __builtin_fma(-h, g, 0.5);
    double y2 = __builtin_fma(y1, rr, y1);
    out[6 * i + 0] = fabs(r0 - r_ex) / r_ex; out[6 * i + 1] = fabs(r1 - r_ex) / r_ex; out[6 * i + 2] = fabs(r2 - r_ex) / r_ex;
    out[6 * i + 3] = fabs(y0 - s_ex) / s_ex; out[6 * i + 4] = fabs(y1 - s_ex) / s_ex; out[6 * i + 5] = fabs(y2 - s_ex) / s_ex;
}
int main(){
    int n = 1 << 22; double* d; (void)hipMalloc(&d, sizeof(double) * 6 * n);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, n);
    double* h = (double*)malloc(sizeof(double) * 6 * n); (void)hipMemcpy(h, d, sizeof(double) * 6 * n, hipMemcpyDeviceToHost);
    double m[6] = {0};
    for(int i = 0; i < n; i++) for(int j = 0; j < 6; j++) if(h[6 * i + j] > m[j]) m[j] = h[6 * i + j];
    printf("max rel err: rcp raw %.3e  +1NR %.3e  +2NR %.3e | rsq raw %.3e  +1NR %.3e  +2NR %.3e\n", m[0], m[1], m[2], m[3], m[4], m[5]);
    return 0;
}
