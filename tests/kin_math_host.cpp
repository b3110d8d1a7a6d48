// Test harness (CPU): the per-row arithmetic of the kinematics kernels - ikflow_amd/csrc/kin_math.h, the very source the GPU runs - compiled with
// g++ and driven row by row, so that tests/test_kin_math_host.py can hold it against the oracle without a GPU.  Not part of the product.
#include "../ikflow_amd/csrc/kin_math.h"

using ikf::Chain;

template <int N>
static void run(const Chain* ch, int what, const float* tgt, const float* q, long long n, float* out, float* out2) {
  for (long long i = 0; i < n; ++i) {
    float qv[N];
    for (int j = 0; j < N; ++j) qv[j] = q[i * N + j];
    if (what == 0) {                       // FK -> [n x 7]
      ikf::fk_pose_f32<N>(ch, qv, out + i * 7);
    } else if (what == 1) {                // pose error -> pos [n], rot [n]
      ikf::pose_error_f32<N>(ch, qv, tgt + i * 7, out + i, out2 + i);
    } else {                               // LM step, fp32 (2) or fp64 inside (3) -> [n x N]
      if (what == 2) ikf::lm_step_row<N, float>(ch, tgt + i * 7, qv);
      else ikf::lm_step_row<N, double>(ch, tgt + i * 7, qv);
      for (int j = 0; j < N; ++j) out[i * N + j] = qv[j];
    }
  }
}

extern "C" int kin_math_host(const void* chain, int what, const float* tgt, const float* q, long long n, float* out, float* out2) {
  const Chain* ch = static_cast<const Chain*>(chain);
  switch (ch->ndof) {
    case 6: run<6>(ch, what, tgt, q, n, out, out2); return 0;
    case 7: run<7>(ch, what, tgt, q, n, out, out2); return 0;
    case 8: run<8>(ch, what, tgt, q, n, out, out2); return 0;
    default: return 1;
  }
}
extern "C" int kin_math_chain_bytes() { return (int)sizeof(Chain); }
