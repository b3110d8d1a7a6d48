// The row-owner flow kernel (ikflow_amd/csrc/flow_rowowner.hip) stand-alone: every CU streams every weight while its 16 rows stay on chip.
// Random nn.Linear-default weights in the engine's arena layouts -> device pack -> one launch per call; checks 32 rows against an fp64
// host reference of the inverse pass; prints ms per call, shader cycles per subnet (median over the workgroups) against the
// 2 x 131,072-cycle matrix-pipe floor, and the spread over workgroups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/rowowner_probe.hip -o tools/bin/rowowner_probe
//   tools/bin/rowowner_probe [rows=4096] [iters=20] [nbuf=4] [blocks=12] [D=7] [G=1] [duo=0]
// G = 2 / 4 / 8: the cluster form (k_flow_cluster<G>: G workgroups per 16-row tile split the hidden columns and exchange activations
// through global memory) - rows * G / 16 must not exceed the CU count.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../ikflow_amd/csrc/flow_rowowner.hip"
#include "flow_duo_probe.inc"   // the half-CU cluster form: measured and dropped, kept as a probe
#include "flow_pair_probe.inc"  // two row tiles per workgroup in lockstep (r05): measured and dropped, kept as a probe

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
using namespace ikf;

struct HostSubnet {
  int n_x, n_out, which, block;
  std::vector<float> w1t, wsoft, b1, w2, b2, w3, b3, wl, bl;  // arena layouts: w1t [n_x + 7][W]; w2 / w3 [W][W]; wl [n_out][W]
};

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096;
  const int iters = argc > 2 ? atoi(argv[2]) : 20;
  const int nbuf = argc > 3 ? atoi(argv[3]) : 4;
  const int NB = argc > 4 ? atoi(argv[4]) : 12;
  const int D = argc > 5 ? atoi(argv[5]) : 7;
  const int G = argc > 6 ? atoi(argv[6]) : 1;
  const int duo = argc > 7 ? atoi(argv[7]) : 0;   // 1: the half-CU form (k_flow_duo<G>, two workgroups per CU); 2: the XCD-local cluster form (G = 8 / 16)
  const int L1 = D / 2, L2 = D - L1, W = RO_W, ndof = 7;
  const int n_sub = 2 * NB;
  std::mt19937 rng(7);
  auto uni = [&](float b) { return std::uniform_real_distribution<float>(-b, b)(rng); };
  std::vector<HostSubnet> subs(n_sub);
  std::vector<std::vector<int>> perm_inv(NB, std::vector<int>(D));
  for (int b = 0; b < NB; ++b) {
    std::vector<int> p(D);
    for (int i = 0; i < D; ++i) p[i] = i;
    std::shuffle(p.begin(), p.end(), rng);
    for (int i = 0; i < D; ++i) perm_inv[b][p[i]] = i;
  }
  for (int s = 0; s < n_sub; ++s) {
    HostSubnet& h = subs[s];
    h.block = NB - 1 - s / 2; h.which = 1 + (s & 1);
    h.n_x = h.which == 1 ? L1 : L2; h.n_out = 2 * (h.which == 1 ? L2 : L1);
    const int in = h.n_x + 8;
    const float bi = 1.f / sqrtf((float)in), bw = 1.f / sqrtf((float)W);
    h.w1t.resize((size_t)(h.n_x + 7) * W); h.wsoft.resize(W); h.b1.resize(W);
    for (auto& v : h.w1t) v = uni(bi);
    for (auto& v : h.wsoft) v = uni(bi);
    for (auto& v : h.b1) v = uni(bi);
    h.w2.resize((size_t)W * W); h.w3.resize((size_t)W * W); h.b2.resize(W); h.b3.resize(W);
    for (auto& v : h.w2) v = uni(bw);
    for (auto& v : h.w3) v = uni(bw);
    for (auto& v : h.b2) v = uni(bw);
    for (auto& v : h.b3) v = uni(bw);
    h.wl.resize((size_t)h.n_out * W); h.bl.resize(h.n_out);
    for (auto& v : h.wl) v = uni(bw);
    for (auto& v : h.bl) v = uni(bw);
  }
  // device: arena pieces per subnet, stream image
  float* d_stream = nullptr;
  const size_t stream_floats = rowowner_stream_floats(n_sub);
  CK(hipMalloc(&d_stream, stream_floats * 4));
  CK(hipMemset(d_stream, 0, stream_floats * 4));
  auto up = [&](const std::vector<float>& v) { float* d; CK(hipMalloc(&d, v.size() * 4)); CK(hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice)); return d; };
  std::vector<RoSubnet> hsub(n_sub);
  for (int s = 0; s < n_sub; ++s) {
    const HostSubnet& h = subs[s];
    SubnetWeights w{};
    float *a1 = up(h.w1t), *a2 = up(h.wsoft), *a3 = up(h.b1), *a4 = up(h.w2), *a5 = up(h.b2), *a6 = up(h.w3), *a7 = up(h.b3), *a8 = up(h.wl), *a9 = up(h.bl);
    w.w_first_t = a1; w.w_soft = a2; w.b_first = a3; w.w_mid[0] = a4; w.b_mid[0] = a5; w.w_mid[1] = a6; w.b_mid[1] = a7; w.w_last = a8; w.b_last = a9;
    w.n_x = h.n_x; w.n_out = h.n_out;
    CK(launch_rowowner_pack(w, d_stream + (size_t)s * rowowner_subnet_floats(), nullptr));
    CK(hipDeviceSynchronize());
    for (float* p : {a1, a2, a3, a4, a5, a6, a7, a8, a9}) CK(hipFree(p));
    RoSubnet& r = hsub[s];
    memset(&r, 0, sizeof(r));
    for (int o = 0; o < h.n_out; ++o) r.b_last[o] = h.bl[o];
    for (int d = 0; d < 16; ++d) r.perm_inv[d] = d < D ? perm_inv[h.block][d] : d;
    r.which = h.which; r.n_x = h.n_x; r.x_off = h.which == 1 ? 0 : L1; r.n_half = h.n_out / 2;
  }
  RoSubnet* d_sub; CK(hipMalloc(&d_sub, sizeof(RoSubnet) * n_sub)); CK(hipMemcpy(d_sub, hsub.data(), sizeof(RoSubnet) * n_sub, hipMemcpyHostToDevice));
  // inputs
  std::vector<float> hx((size_t)M * D), hp((size_t)M * 7), Minv((size_t)D * D, 0.f), blin(D, 0.f), lo(ndof, -2.8f), hi(ndof, 2.8f);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& v : hx) v = nd(rng);
  for (int r = 0; r < M; ++r) {
    float q[4], n = 0;
    for (int k = 0; k < 3; ++k) hp[(size_t)r * 7 + k] = uni(0.8f);
    for (int k = 0; k < 4; ++k) { q[k] = nd(rng); n += q[k] * q[k]; }
    for (int k = 0; k < 4; ++k) hp[(size_t)r * 7 + 3 + k] = q[k] / sqrtf(n);
  }
  for (int k = 0; k < D; ++k) Minv[(size_t)k * D + k] = k < ndof ? 2.9f : 1.f;
  float *d_x = up(hx), *d_p = up(hp), *d_Minv = up(Minv), *d_blin = up(blin), *d_lo = up(lo), *d_hi = up(hi), *d_q;
  CK(hipMalloc(&d_q, (size_t)M * ndof * 4));
  const unsigned n_tiles = (M + RO_ROWS - 1) / RO_ROWS;
  const unsigned grid = duo == 3 ? pair_grid((int)n_tiles, G) : ((duo == 2 || (duo == 4 && (G == 4 || G == 8 || G == 16))) ? (n_tiles + 7) / 8 * 8 : n_tiles) * (unsigned)G;   // (the XCD-local form pads to whole groups of 8 row tiles)
  unsigned long long* d_trace; CK(hipMalloc(&d_trace, (size_t)grid * 64 * 8)); CK(hipMemset(d_trace, 0, (size_t)grid * 64 * 8));
  RoArgs a{};
  a.stream = d_stream; a.stream_bytes = (unsigned)(stream_floats * 4); a.sub = d_sub; a.n_sub = n_sub; a.x0 = d_x;
  a.ps = PoseSource{d_p, nullptr, (long long)M, 7, 0.f}; a.row0 = 0; a.M = M; a.D = D; a.L1 = L1; a.ndof = ndof; a.clamp = 2.5f; a.slope = 0.01f;
  a.M_inv = d_Minv; a.b_lin = d_blin; a.lo = d_lo; a.hi = d_hi; a.clamp_limits = 1; a.sigmoid = 0; a.q_out = d_q; a.trace = nullptr;
  RcArgs rc{};
  int* h_give_up = nullptr;
  if (G > 1) {
    rc.n_rt = (M + RO_ROWS - 1) / RO_ROWS;
    const int n_rt2 = (rc.n_rt + 1) / 2 * 2;   // (the pair form's buffers hold an even number of row tiles)
    CK(hipMalloc(&rc.xbuf, cluster_xbuf_floats(n_rt2) * 4));
    CK(hipMalloc(&rc.pbuf, cluster_sync_bytes(n_rt2, G)));
    const int n_rtb = duo == 3 ? n_rt2 : rc.n_rt;
    rc.flags = reinterpret_cast<unsigned*>(rc.pbuf + (size_t)n_rtb * G * 256);
    rc.abort_word = rc.flags + (size_t)n_rtb * G * 32;
    CK(hipHostMalloc(&h_give_up, 4, hipHostMallocMapped)); *h_give_up = 0;
    rc.give_up = h_give_up;
    if (duo == 4 || duo == 5)   // the tagged hand-over: buffers created as 0xff bytes, nothing zeroed per launch
      CK(cluster_tagged_init(rc.xbuf, cluster_xbuf_floats(n_rt2), rc.pbuf, cluster_sync_bytes(n_rt2, G) - 128, rc.abort_word, nullptr));
    CK(hipMalloc(&rc.xcc_words, (size_t)(n_rt2 + 8) * 32 * 4));   // (r06: the tagged XCD-local form's placement words, one line per row tile)
    CK(hipMemset(rc.xcc_words, 0xff, (size_t)(n_rt2 + 8) * 32 * 4));
    rc.launch_seq = 1;
  }
  auto launch = [&]() -> hipError_t {
    if (G > 1) { rc.ro = a; if (duo == 4 || duo == 5) return launch_flow_cluster_tagged(rc, G, nullptr, 0, /*local=*/duo == 4); return duo == 3 ? launch_flow_pair(rc, G, nullptr) : duo == 1 ? launch_flow_duo(rc, G, nullptr) : launch_flow_cluster(rc, G, nullptr, 0, /*local=*/duo == 2); }
    return launch_flow_rowowner(a, nbuf, nullptr);
  };
#define launch_flow_rowowner(a_, n_, s_) launch()
  CK(launch_flow_rowowner(a, nbuf, nullptr));
  CK(hipDeviceSynchronize());
  if (h_give_up && *h_give_up) { printf("GIVE UP: a wait ran out\n"); return 3; }
  std::vector<float> hq((size_t)M * ndof);
  CK(hipMemcpy(hq.data(), d_q, hq.size() * 4, hipMemcpyDeviceToHost));
  // fp64 reference on the first and the last 16 rows
  double max_err = 0;
  std::vector<int> rows;
  for (int r = 0; r < 16 && r < M; ++r) rows.push_back(r);
  for (int r = std::max(16, M - 16); r < M; ++r) rows.push_back(r);
  for (int r : rows) {
    std::vector<double> x(D);
    for (int d = 0; d < D; ++d) x[d] = hx[(size_t)r * D + d];
    std::vector<double> h1(W), h2(W), u(16);
    for (int s = 0; s < n_sub; ++s) {
      const HostSubnet& h = subs[s];
      const int off_in = h.which == 1 ? 0 : L1, off_out = h.which == 1 ? L1 : 0, nl = h.n_out / 2;
      for (int k = 0; k < h.n_x; ++k) u[k] = x[off_in + k];
      for (int k = 0; k < 7; ++k) u[h.n_x + k] = hp[(size_t)r * 7 + k];
      for (int c = 0; c < W; ++c) {
        double acc = h.b1[c];
        for (int k = 0; k < h.n_x + 7; ++k) acc += u[k] * h.w1t[(size_t)k * W + c];
        h1[c] = acc > 0 ? acc : 0.01 * acc;
      }
      for (int c = 0; c < W; ++c) {
        double acc = h.b2[c];
        const float* wr = &h.w2[(size_t)c * W];
        for (int k = 0; k < W; ++k) acc += h1[k] * wr[k];
        h2[c] = acc > 0 ? acc : 0.01 * acc;
      }
      for (int c = 0; c < W; ++c) {
        double acc = h.b3[c];
        const float* wr = &h.w3[(size_t)c * W];
        for (int k = 0; k < W; ++k) acc += h2[k] * wr[k];
        h1[c] = acc > 0 ? acc : 0.01 * acc;
      }
      std::vector<double> o(h.n_out);
      for (int j = 0; j < h.n_out; ++j) {
        double acc = h.bl[j];
        for (int k = 0; k < W; ++k) acc += h1[k] * h.wl[(size_t)j * W + k];
        o[j] = acc;
      }
      for (int j = 0; j < nl; ++j) x[off_out + j] = (x[off_out + j] - o[nl + j]) * exp(-2.5 * 0.636 * atan(o[j]));
      if (h.which == 2) {
        std::vector<double> y(D);
        for (int d = 0; d < D; ++d) y[d] = x[perm_inv[h.block][d]];
        x = y;
      }
    }
    for (int j = 0; j < ndof; ++j) {
      double q = x[j] * 2.9;
      q = std::min(std::max(q, -2.8), 2.8);
      max_err = std::max(max_err, fabs(q - (double)hq[(size_t)r * ndof + j]));
    }
  }
  printf("%sG %d rows %d blocks %d D %d nbuf %d: max |q - fp64 reference| over %zu rows = %.3g %s\n", duo == 3 ? "pair " : duo == 4 ? "tagged local " : duo == 5 ? "tagged spread " : duo ? "duo " : "", G, M, NB, D, nbuf, rows.size(), max_err, max_err < 2e-5 ? "OK" : "MISMATCH");
  // timing
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) CK(launch_flow_rowowner(a, nbuf, nullptr));
  CK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters; ++i) CK(launch_flow_rowowner(a, nbuf, nullptr));
  CK(hipEventRecord(e1, nullptr));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= iters;
  const double flop = 2.0 * M * (double)n_sub * 2 * W * W;
  printf("%.4f ms per call = %.3f M rows/s; hidden contractions alone %.1f TFLOP/s (%.3f of 157.3)\n", ms, M / ms * 1e-3, flop / ms * 1e-9, flop / ms * 1e-9 / 157.3);
  {  // the same launches, each bracketed by its own event pair (what ikf_profile_begin/_end does)
    std::vector<hipEvent_t> ev(2 * iters);
    for (auto& e : ev) CK(hipEventCreate(&e));
    for (int i = 0; i < iters; ++i) {
      CK(hipEventRecord(ev[2 * i], nullptr));
      CK(launch_flow_rowowner(a, nbuf, nullptr));
      CK(hipEventRecord(ev[2 * i + 1], nullptr));
    }
    CK(hipDeviceSynchronize());
    double sum = 0, span = 0;
    for (int i = 0; i < iters; ++i) { float m1; CK(hipEventElapsedTime(&m1, ev[2 * i], ev[2 * i + 1])); sum += m1; }
    { float m1; CK(hipEventElapsedTime(&m1, ev[0], ev[2 * iters - 1])); span = m1; }
    printf("per-launch event pairs: mean %.4f ms per launch; first event -> last event %.4f ms per launch\n", sum / iters, span / iters);
  }
  // in-kernel timeline
  a.trace = d_trace;
  for (int i = 0; i < 3; ++i) CK(launch_flow_rowowner(a, nbuf, nullptr));
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> tr((size_t)grid * 64);
  CK(hipMemcpy(tr.data(), d_trace, tr.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> per_sub, total, start;
  unsigned long long t_min = ~0ull;
  for (unsigned b = 0; b < grid; ++b) t_min = std::min(t_min, tr[(size_t)b * 64]);
  for (unsigned b = 0; b < grid; ++b) {
    const unsigned long long* p = &tr[(size_t)b * 64];
    const int ns = std::min(n_sub, 31);
    for (int s = 1; s < ns; ++s) per_sub.push_back((double)(p[1 + s] - p[s]));
    total.push_back((double)(p[34] - p[0]));
    start.push_back((double)(p[0] - t_min));
  }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
  auto mx = [](const std::vector<double>& v) { return v.empty() ? 0.0 : *std::max_element(v.begin(), v.end()); };
  auto mn = [](const std::vector<double>& v) { return v.empty() ? 0.0 : *std::min_element(v.begin(), v.end()); };
  printf("cycles per subnet: median %.0f min %.0f max %.0f (matrix-pipe floor 2 x 131072 + 2 x 2048 = 266240); per workgroup total: median %.0f max %.0f; start skew max %.0f\n",
         med(per_sub), mn(per_sub), mx(per_sub), med(total), mx(total), mx(start));
  {
    const unsigned long long* p0 = &tr[0];
    const double us = (double)(p0[36] - p0[35]) / 100.0;  // 100 MHz constant clock
    printf("workgroup 0 of the last of three back-to-back traced launches: %.1f us, %.0f shader cycles -> %.3f GHz\n", us, (double)(p0[34] - p0[0]), (double)(p0[34] - p0[0]) / us * 1e-3);
  }
  if (grid >= 8) {  // workgroup b runs on XCD b % 8 (observed placement): wall time and effective clock per XCD
    unsigned long long w0 = ~0ull, w1 = 0;
    for (unsigned b = 0; b < grid; ++b) { w0 = std::min(w0, tr[(size_t)b * 64 + 35]); w1 = std::max(w1, tr[(size_t)b * 64 + 36]); }
    printf("first workgroup start -> last workgroup end: %.1f us\n", (double)(w1 - w0) / 100.0);
    for (int x = 0; x < 8; ++x) {
      double us = 0, cyc = 0, st = 0, en = 0; int n = 0;
      for (unsigned b = x; b < grid && b < 256; b += 8) {
        const unsigned long long* p = &tr[(size_t)b * 64];
        us += (double)(p[36] - p[35]) / 100.0; cyc += (double)(p[34] - p[0]); st += (double)(p[35] - w0) / 100.0; en += (double)(p[36] - w0) / 100.0; ++n;
      }
      printf("  XCD %d: %d workgroups, mean %.1f us, %.3f GHz, mean start +%.1f us, mean end +%.1f us\n", x, n, us / n, cyc / us * 1e-3, st / n, en / n);
    }
  }
#ifdef IKF_RC_FINE_STAMPS
  if (n_sub > 2 && (duo == 4 || duo == 5)) {   // the tagged h2 gather of subnet 2 taken apart: own-slice k groups, the wait for the peers, the rest of hidden 3
    std::vector<double> own, wait, rest, tries;
    for (unsigned g = 0; g < grid; ++g) {
      const unsigned long long* p = &tr[(size_t)g * 64];
      if (p[46] == 0 || p[47] == 0) continue;
      own.push_back((double)(p[46] - p[42])); wait.push_back((double)(p[47] - p[46])); rest.push_back((double)(p[43] - p[47])); tries.push_back((double)p[50] / (3.0 * n_sub));
    }
    printf("  hidden 3 of subnet 2: own-slice k groups median %.0f max %.0f | h2 gather (issue -> all peers in LDS) median %.0f min %.0f max %.0f | remaining k groups + epilogue median %.0f | failed passes per gather median %.2f max %.2f\n",
           med(own), mx(own), med(wait), mn(wait), mx(wait), med(rest), med(tries), mx(tries));
  }
#endif
  if (n_sub > 2 && duo == 3) {   // pair form: the four slots of subnet 2 (compute wave 0) and the comm team's four jobs (comm wave 0)
    const char* cn[4] = {"A.h2", "B.h2", "A.h3 + last", "B.h3 + last"};
    for (int k = 0; k < 4; ++k) {
      std::vector<double> w, b;
      for (unsigned g = 0; g < grid; ++g) {
        const unsigned long long* p = &tr[(size_t)g * 64];
        w.push_back((double)(p[41 + 2 * k] - p[40 + 2 * k]));
        if (k < 3) b.push_back((double)(p[42 + 2 * k] - p[41 + 2 * k]));
      }
      printf("  compute slot %-12s work median %7.0f max %7.0f   then barrier wait median %6.0f max %6.0f\n", cn[k], med(w), mx(w), med(b), mx(b));
    }
    const char* mn_[4] = {"A: sums, coupling, first Linear", "B: sums, coupling, first Linear", "A: h2 out / in", "B: h2 out / in"};
    for (int k = 0; k < 4; ++k) {
      std::vector<double> v1, v2, v3;
      for (unsigned g = 0; g < grid; ++g) {
        const unsigned long long* p = &tr[(size_t)g * 64 + 48 + 4 * k];
        v1.push_back((double)(p[1] - p[0])); v2.push_back((double)(p[2] - p[1])); v3.push_back((double)(p[3] - p[2]));
      }
      printf("  comm job %-34s publish %6.0f (max %6.0f)  poll %6.0f (max %6.0f)  rest %6.0f (max %6.0f)\n", mn_[k], med(v1), mx(v1), med(v2), mx(v2), med(v3), mx(v3));
    }
#ifdef RP_FINE_STAMPS
    {
      std::vector<double> v1, v2, v3, v4;
      const int b0 = 56 + 4 * (RP_FINE_STAMPS - 1);
      for (unsigned g = 0; g < grid; ++g) {
        const unsigned long long* p = &tr[(size_t)g * 64];
        v1.push_back((double)(p[39] - p[b0])); v2.push_back((double)(p[37] - p[39])); v3.push_back((double)(p[38] - p[37])); v4.push_back((double)(p[b0 + 1] - p[38]));
      }
      printf("  X2 publisher of tile %d: vmcnt(0) at entry %6.0f (max %6.0f), LDS reads %6.0f (max %6.0f), store issue + drain %6.0f (max %6.0f), flag store + stamp %6.0f (max %6.0f)\n", RP_FINE_STAMPS - 1, med(v1), mx(v1), med(v2), mx(v2), med(v3), mx(v3), med(v4), mx(v4));
    }
#endif
  } else if (n_sub > 2) {
    const char* names[5] = {"first Linear", "hidden 2", "hidden 3", "last Linear", "coupling + next input rows"};
    for (int ph = 0; ph < 5; ++ph) {
      std::vector<double> v;
      for (unsigned b = 0; b < grid; ++b) v.push_back((double)(tr[(size_t)b * 64 + 41 + ph] - tr[(size_t)b * 64 + 40 + ph]));
      printf("  phase %-28s median %8.0f max %8.0f cycles\n", names[ph], med(v), mx(v));
    }
  }
  return max_err < 2e-5 ? 0 : 2;
}
