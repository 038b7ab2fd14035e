// Stand-alone check of csrc/fft_half_f64.hpp on the GPU (built and run by tests/test_gpu_torus_fft.py): the two half transforms of a
// random polynomial against the definition evaluated in long double on the host, and the inverse halves on S / D built from them
// (the round trip returns the polynomial).  Prints "ok <max forward error> <max round-trip error>" or a line starting with "FAIL".
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "../../bounty-matrix-inversion_amd/csrc/fft_half_f64.hpp"

using namespace ffth;

__global__ void k_fwd(const double *a, const double *g_tw, double2 *out) {   // 2 wavefronts: h = wave
    __shared__ double tw[HT_WORDS];
    for (int i = threadIdx.x; i < HT_WORDS; i += blockDim.x) tw[i] = g_tw[i];
    __syncthreads();
    const int h = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double re[4], im[4];
    for (int r = 0; r < 4; r++) {
        re[r] = a[2 * (lane + 64 * r) + h];
        im[r] = a[2 * (lane + 64 * r) + h + 512];
    }
    C v[4];
    if (h) forward_half<1>(re, im, v, lane, tw);
    else forward_half<0>(re, im, v, lane, tw);
    for (int r = 0; r < 4; r++) out[h * 256 + r * 64 + lane] = double2{v[r].r, v[r].i};
}

__global__ void k_inv(const double2 *sd, const double *g_tw, double *a) {   // sd: [h][slot]
    __shared__ double tw[HT_WORDS];
    for (int i = threadIdx.x; i < HT_WORDS; i += blockDim.x) tw[i] = g_tw[i];
    __syncthreads();
    const int h = threadIdx.x >> 6, lane = threadIdx.x & 63;
    C v[4];
    for (int r = 0; r < 4; r++) {
        const double2 t = sd[h * 256 + r * 64 + lane];
        v[r] = C{t.x, t.y};
    }
    double re[4], im[4];
    if (h) inverse_half<1>(v, re, im, lane, tw);
    else inverse_half<0>(v, re, im, lane, tw);
    for (int r = 0; r < 4; r++) {
        a[2 * (lane + 64 * r) + h] = re[r];
        a[2 * (lane + 64 * r) + h + 512] = im[r];
    }
}

#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("FAIL hip %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)

int main() {
    std::vector<double> tw(HT_WORDS), a(N), back(N);
    build_tables(tw.data());
    unsigned long long s = 12345;
    for (int i = 0; i < N; i++) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; a[i] = (double)((long long)(s >> 44) % 1024 - 512); }
    double *d_a, *d_tw, *d_back;
    double2 *d_out, *d_sd;
    OK(hipMalloc(&d_a, N * 8)); OK(hipMalloc(&d_tw, HT_WORDS * 8)); OK(hipMalloc(&d_back, N * 8));
    OK(hipMalloc(&d_out, 512 * 16)); OK(hipMalloc(&d_sd, 512 * 16));
    OK(hipMemcpy(d_a, a.data(), N * 8, hipMemcpyHostToDevice));
    OK(hipMemcpy(d_tw, tw.data(), HT_WORDS * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fwd, dim3(1), dim3(128), 0, 0, d_a, d_tw, d_out);
    OK(hipDeviceSynchronize());
    std::vector<double2> out(512), sd(512);
    OK(hipMemcpy(out.data(), d_out, 512 * 16, hipMemcpyDeviceToHost));
    const long double PI = 3.14159265358979323846264338327950288L;
    double ferr = 0;
    std::vector<long double> Fr(512), Fi(512);
    for (int k = 0; k < 512; k++) {
        long double sr = 0, si = 0;
        for (int j = 0; j < 512; j++) {
            const long double ang = PI * (long double)((j * (4 * k + 1)) % 2048) / 1024.0L;
            const long double c = cosl(ang), sn = sinl(ang);
            sr += a[j] * c - a[j + 512] * sn;
            si += a[j] * sn + a[j + 512] * c;
        }
        Fr[k] = sr; Fi[k] = si;
    }
    for (int p = 0; p < 256; p++) {
        const int k = slot_freq(p >> 6, p & 63);
        const double lo_r = out[p].x + out[256 + p].x, lo_i = out[p].y + out[256 + p].y;
        const double hi_r = out[p].x - out[256 + p].x, hi_i = out[p].y - out[256 + p].y;
        ferr = fmax(ferr, fmax(fabs(lo_r - (double)Fr[k]), fabs(lo_i - (double)Fi[k])));
        ferr = fmax(ferr, fmax(fabs(hi_r - (double)Fr[k + 256]), fabs(hi_i - (double)Fi[k + 256])));
        // S and D of the identity product Y = F
        const long double ang = PI * (long double)((4 * k) % 2048) / 1024.0L;
        const double wr = (double)cosl(ang), wi = (double)sinl(ang);
        sd[p] = double2{lo_r + hi_r, lo_i + hi_i};
        const double dr = lo_r - hi_r, di = lo_i - hi_i;
        sd[256 + p] = double2{dr * wr + di * wi, di * wr - dr * wi};
    }
    OK(hipMemcpy(d_sd, sd.data(), 512 * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_inv, dim3(1), dim3(128), 0, 0, d_sd, d_tw, d_back);
    OK(hipDeviceSynchronize());
    OK(hipMemcpy(back.data(), d_back, N * 8, hipMemcpyDeviceToHost));
    double rerr = 0;
    for (int i = 0; i < N; i++) rerr = fmax(rerr, fabs(back[i] - a[i]));
    if (ferr > 1e-6 || rerr > 1e-6) { printf("FAIL forward %.3g round trip %.3g\n", ferr, rerr); return 1; }
    printf("ok %.3g %.3g\n", ferr, rerr);
    return 0;
}
