// Dev harness: the small-S xSlot kernels with -DXS16_TIMING; prints the cycle deltas (s_memtime) between the
// XS16_STAMP() points of wave 0, averaged over the images, for the forward and the backward launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DXS16_TIMING tools_dev/xs_small_timing.hip -o /tmp/xs16 && /tmp/xs16 [B S N T L]
#include "../scouter_amd/csrc/xslot_fwd.hip"
#include "../scouter_amd/csrc/xslot_bwd.hip"
#include <stdarg.h>
#include <vector>
void sc_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }
int sc_check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : -2; }
ScProfScope::ScProfScope(const char*, hipStream_t st, double, double) : stream(st), slot(-1) {}
ScProfScope::~ScProfScope() {}

static float* dev_rand(size_t n, float scale, bool positive) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) { float u = (float)rand() / RAND_MAX - (positive ? 0.f : 0.5f); h[i] = u * scale; }
    float* d; hipMalloc(&d, n * 4 + 16); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d;
}
static void report(const char* what, long long* dst, int B) {
    std::vector<long long> st((size_t)B * 64);
    hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost);
    int n = 0; while (n < 64 && st[n] != 0) ++n;
    printf("%s: %d stamps; mean delta (cycles) between consecutive stamps over %d images:\n", what, n, B);
    for (int k = 1; k < n; ++k) {
        double s = 0; for (int b = 0; b < B; ++b) s += (double)(st[b * 64 + k] - st[b * 64 + k - 1]);
        printf("  %2d -> %2d : %8.0f\n", k - 1, k, s / B);
    }
    double tot = 0; for (int b = 0; b < B; ++b) tot += (double)(st[b * 64 + n - 1] - st[b * 64]);
    printf("  total    : %8.0f cycles\n", tot / B);
}
int main(int argc, char** argv) {
    int B = argc > 1 ? atoi(argv[1]) : 70, S = argc > 2 ? atoi(argv[2]) : 10, N = argc > 3 ? atoi(argv[3]) : 49,
        T = argc > 4 ? atoi(argv[4]) : 3, L = argc > 5 ? atoi(argv[5]) : 3;
    const int d = 64, spc = 1;
    float* X = dev_rand((size_t)B * N * d, 1.f, true); float* PE = dev_rand(N * d, 0.3f, false);
    const float* tw[8]; const float* tb[8];
    for (int l = 0; l < L; ++l) { tw[l] = dev_rand(d * d, 0.2f, false); tb[l] = dev_rand(d, 0.1f, false); }
    float* s0 = dev_rand(S * d, 0.5f, true);
    float* w_ih = dev_rand(192 * d, 0.2f, false); float* w_hh = dev_rand(192 * d, 0.2f, false);
    float* b_ih = dev_rand(192, 0.1f, false); float* b_hh = dev_rand(192, 0.1f, false);
    float* logits = dev_rand((size_t)B * S, 0, false); float* attn = dev_rand((size_t)B * S * N, 0, false);
    float* area = dev_rand(B, 0, false); float* Ks = dev_rand((size_t)B * N * d, 0, false);
    float* Hs = dev_rand((size_t)L * B * N * d, 0, false); float* states = dev_rand((size_t)T * B * S * d, 0, false);
    float* dlog = dev_rand((size_t)B * S, 0.01f, false); float* ga = dev_rand(1, 1e-4f, true);
    float* dX = dev_rand((size_t)B * N * d, 0, false); float* dgi = dev_rand((size_t)T * B * S * 192, 0, false);
    float* dgh = dev_rand((size_t)T * B * S * 192, 0, false); float* Us = dev_rand((size_t)T * B * S * d, 0, false);
    float* ds0 = dev_rand((size_t)B * S * d, 0, false); float* dZ = dev_rand((size_t)L * B * N * d, 0, false);
    const size_t wsb = scouter_xslot_bwd_workspace_bytes(B, N, d, S, T);
    void* ws; hipMalloc(&ws, wsb + 16);
    long long* st_d; hipMalloc(&st_d, (size_t)B * 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; ++pass) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(st_d, 0, (size_t)B * 64 * 8);
            hipMemcpyToSymbol(HIP_SYMBOL(g_xs16_stamps), &st_d, sizeof(st_d));
            hipDeviceSynchronize();
            hipEventRecord(e0);
            int rc = pass == 0 ? scouter_xslot_fwd_f32(X, PE, tw, tb, s0, w_ih, w_hh, b_ih, b_hh, B, N, d, S, spc, T, L, 1.f, logits, attn,
                                                       area, Ks, Hs, states, nullptr)
                               : scouter_xslot_bwd_f32(X, PE, tw, s0, w_ih, w_hh, b_ih, b_hh, Ks, Hs, states, dlog, ga, B, N, d, S, spc, T,
                                                       L, 1.f, dX, dgi, dgh, Us, ds0, dZ, ws, wsb, nullptr);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s launch %d: %.1f us rc=%d (%s)\n", pass ? "bwd" : "fwd", rep, ms * 1e3, rc, hipGetErrorString(hipGetLastError()));
        }
        report(pass ? "backward" : "forward", st_d, B);
    }
    return 0;
}
