// Dev harness: compiles the xSlot backward kernel with -DXS_TIMING and prints the cycle-counter deltas between the
// XSB_STAMP() points of wave 0, averaged over the images.   hipcc --offload-arch=gfx950 -O3 -DXS_TIMING -I scouter_amd/csrc
//   tools_dev/xs_bwd_phase_timing.hip -o /tmp/xsb_timing && /tmp/xsb_timing [B S N T]
// Per iteration: A(r, tau) | B1 of tile 0: D/A, U, GRU (t < T-1), dA/G | rest of B1 + barrier | B2 + barrier.
#include "../scouter_amd/csrc/xslot_bwd.hip"
#include <stdarg.h>
#include <stdlib.h>
#include <vector>
void sc_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }
int sc_check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : -2; }
ScProfScope::ScProfScope(const char*, hipStream_t st, double, double) : stream(st), slot(-1) {}
ScProfScope::~ScProfScope() {}

static float* dev_rand(size_t n, float scale, bool positive) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) { float u = (float)rand() / RAND_MAX - (positive ? 0.f : 0.5f); h[i] = u * scale; }
    float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d;
}
int main(int argc, char** argv) {
    int B = argc > 1 ? atoi(argv[1]) : 256, S = argc > 2 ? atoi(argv[2]) : 300, N = argc > 3 ? atoi(argv[3]) : 49,
        T = argc > 4 ? atoi(argv[4]) : 3;
    const int d = 64, spc = 1, L = 1;
    XsBwdArgs a{};
    a.X = dev_rand((size_t)B * N * d, 1.f, true); a.PE = dev_rand(N * d, 0.3f, false);
    a.tok_w[0] = dev_rand(d * d, 0.2f, false);
    a.slots0 = dev_rand(S * d, 0.5f, true);
    a.w_ih = dev_rand(192 * d, 0.2f, false); a.w_hh = dev_rand(192 * d, 0.2f, false);
    a.b_ih = dev_rand(192, 0.1f, false); a.b_hh = dev_rand(192, 0.1f, false);
    a.Ksave = dev_rand((size_t)B * N * d, 1.f, true); a.Hsave = dev_rand((size_t)L * B * N * d, 1.f, false);
    a.states = dev_rand((size_t)T * B * S * d, 0.5f, true); a.dlogits = dev_rand((size_t)B * S, 1.f, false);
    a.g_area_sum = dev_rand(1, 1.f, false);
    a.dX = dev_rand((size_t)B * N * d, 0, false); a.dgi = dev_rand((size_t)T * B * S * 192, 0, false);
    a.dgh = dev_rand((size_t)T * B * S * 192, 0, false); a.Usave = dev_rand((size_t)T * B * S * d, 0, false);
    a.ds0 = dev_rand((size_t)B * S * d, 0, false); a.dZ = dev_rand((size_t)L * B * N * d, 0, false);
    size_t wsb = scouter_xslot_bwd_workspace_bytes(B, N, d, S, T);
    hipMalloc(&a.ws, wsb); hipMemset(a.ws, 0, wsb);
    a.B = B; a.N = N; a.S = S; a.C = S / spc; a.spc = spc; a.T = T; a.L = L; a.loss_status = 1.f; a.halves = getenv("HALVES") ? atoi(getenv("HALVES")) : 1;
    hipMalloc(&a.stamps, (size_t)B * 64 * 8); hipMemset(a.stamps, 0, (size_t)B * 64 * 8);
    const int NJT = (N + 31) / 32;
    const size_t lds = xs_bwd_lds_bytes(NJT);
    if (NJT != 2) { printf("harness is specialised for NJT=2\n"); return 1; }
    auto kern = xslot_bwd_kernel<2>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, 0, a);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("launch %d: %.1f us  (%s)\n", rep, ms * 1e3, hipGetErrorString(hipGetLastError()));
    }
    std::vector<long long> st((size_t)B * 64);
    hipMemcpy(st.data(), a.stamps, st.size() * 8, hipMemcpyDeviceToHost);
    int n = 0; while (n < 64 && st[n] != 0) ++n;
    printf("%d stamps; mean delta (counter ticks) between consecutive stamps over %d images:\n", n, B);
    for (int k = 1; k < n; ++k) {
        double s = 0; for (int b = 0; b < B; ++b) s += (double)(st[b * 64 + k] - st[b * 64 + k - 1]);
        printf("  %2d -> %2d : %10.0f\n", k - 1, k, s / B);
    }
    double tot = 0; for (int b = 0; b < B; ++b) tot += (double)(st[b * 64 + n - 1] - st[b * 64]);
    printf("  total    : %10.0f ticks\n", tot / B);
    return 0;
}
