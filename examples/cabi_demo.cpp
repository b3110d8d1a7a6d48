// Stand-alone C++ client of the C-ABI (include/ikflow_amd.h) - no Python, no torch.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 examples/cabi_demo.cpp -Iinclude -Likflow_amd/lib -likflow_amd \
//         -Wl,-rpath,$PWD/ikflow_amd/lib -o cabi_demo
//   ./cabi_demo model.ikfbin out.bin
//
// model.ikfbin (written by tests/test_cabi_cpp.py::write_ikfbin) holds the ikf_model_desc, the state_dict tensors under their
// FrEIA key names, n target poses [n x 7] and n latents [n x D].  The program runs approximate IK, FK of the result
// and the pose error through the library, then the exact-IK entries - ikf_refine_exact on the approximate solutions as seeds,
// and ikf_generate_exact (one round, repeat 1) with a plain C latent callback, which must give the same result - and writes
// [q | fk | pos_err | rot_err | q_refined | valid (as float)] to out.bin.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ikflow_amd.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP: %s\n", hipGetErrorString(e_)); return 2; } } while (0)
#define IKF_OK_(x) do { ikf_status s_ = (x); if (s_ != IKF_OK) { std::fprintf(stderr, "ikf status %d: %s\n", (int)s_, ikf_last_error()); return 3; } } while (0)

template <typename T> static bool rd(FILE* f, T* v, size_t n = 1) { return std::fread(v, sizeof(T), n, f) == n; }

// ikf_latent_fn: the library asks for the latent of retry round `round` ([rows x dim] on the device)
static const float* latent_of_round(void* user, int /*round*/, int64_t /*rows*/, int /*dim*/) { return static_cast<const float*>(user); }

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s model.ikfbin out.bin\n", argv[0]); return 1; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror("open"); return 1; }
  ikf_model_desc desc;
  int32_t n_tensors = 0;
  if (!rd(f, &desc) || !rd(f, &n_tensors)) return 1;
  std::vector<std::string> names(n_tensors);
  std::vector<std::vector<char>> blobs(n_tensors);
  std::vector<ikf_tensor> tensors(n_tensors);
  for (int i = 0; i < n_tensors; ++i) {
    int32_t name_len, dtype, ndim; int64_t shape[4] = {0, 0, 0, 0};
    if (!rd(f, &name_len)) return 1;
    names[i].resize(name_len);
    if (!rd(f, names[i].data(), name_len) || !rd(f, &dtype) || !rd(f, &ndim) || !rd(f, shape, 4)) return 1;
    int64_t numel = 1;
    for (int k = 0; k < ndim; ++k) numel *= shape[k];
    blobs[i].resize((size_t)numel * (dtype == 0 ? 4 : 8));
    if (!rd(f, blobs[i].data(), blobs[i].size())) return 1;
    tensors[i] = ikf_tensor{names[i].c_str(), blobs[i].data(), dtype, ndim, {shape[0], shape[1], shape[2], shape[3]}};
  }
  int64_t n = 0;
  if (!rd(f, &n)) return 1;
  std::vector<float> poses((size_t)n * 7), latent((size_t)n * desc.dim);
  if (!rd(f, poses.data(), poses.size()) || !rd(f, latent.data(), latent.size())) return 1;
  std::fclose(f);

  ikf_model* m = nullptr;
  IKF_OK_(ikf_create(&desc, 0, &m));
  IKF_OK_(ikf_load_weights(m, tensors.data(), n_tensors));
  IKF_OK_(ikf_reserve(m, n));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  float *d_poses, *d_lat, *d_q, *d_fk, *d_pe, *d_re;
  HIP_OK(hipMalloc(&d_poses, poses.size() * 4)); HIP_OK(hipMalloc(&d_lat, latent.size() * 4));
  HIP_OK(hipMalloc(&d_q, (size_t)n * desc.ndof * 4)); HIP_OK(hipMalloc(&d_fk, (size_t)n * 7 * 4));
  HIP_OK(hipMalloc(&d_pe, (size_t)n * 4)); HIP_OK(hipMalloc(&d_re, (size_t)n * 4));
  HIP_OK(hipMemcpyAsync(d_poses, poses.data(), poses.size() * 4, hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(d_lat, latent.data(), latent.size() * 4, hipMemcpyHostToDevice, stream));
  IKF_OK_(ikf_generate_approx(m, d_poses, 0, d_lat, n, 1, 0.0f, d_q, stream));
  IKF_OK_(ikf_forward_kinematics(m, d_q, n, d_fk, stream));
  IKF_OK_(ikf_pose_error(m, d_q, d_poses, n, d_pe, d_re, stream));
  // exact IK, seeds in: refine the approximate solutions (<= 3 LM steps, 5 cm / 0.3 rad so that random weights solve some)
  const float pos_thr = 0.05f, rot_thr = 0.3f;
  float *d_q2, *d_q3; uint8_t *d_v2, *d_v3;
  HIP_OK(hipMalloc(&d_q2, (size_t)n * desc.ndof * 4)); HIP_OK(hipMalloc(&d_q3, (size_t)n * desc.ndof * 4));
  HIP_OK(hipMalloc(&d_v2, (size_t)n)); HIP_OK(hipMalloc(&d_v3, (size_t)n));
  IKF_OK_(ikf_reserve_exact(m, n, 1));
  IKF_OK_(ikf_refine_exact(m, d_poses, n, 1, d_q, 3, pos_thr, rot_thr, d_q2, d_v2, stream));
  // the same through the flow: one retry round with repeat count 1 on the same latent
  const int32_t repeat_counts[1] = {1};
  int64_t stats[4] = {0, 0, 0, 0};
  IKF_OK_(ikf_generate_exact(m, d_poses, n, repeat_counts, 1, 3, pos_thr, rot_thr, latent_of_round, d_lat, d_q3, d_v3, stats, stream));
  std::vector<float> q2((size_t)n * desc.ndof), q3((size_t)n * desc.ndof);
  std::vector<uint8_t> v2(n), v3(n);
  HIP_OK(hipMemcpyAsync(q2.data(), d_q2, q2.size() * 4, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(q3.data(), d_q3, q3.size() * 4, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(v2.data(), d_v2, v2.size(), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(v3.data(), d_v3, v3.size(), hipMemcpyDeviceToHost, stream));
  std::vector<float> q((size_t)n * desc.ndof), fk((size_t)n * 7), pe(n), re(n);
  HIP_OK(hipMemcpyAsync(q.data(), d_q, q.size() * 4, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(fk.data(), d_fk, fk.size() * 4, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(pe.data(), d_pe, pe.size() * 4, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(re.data(), d_re, re.size() * 4, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  FILE* o = std::fopen(argv[2], "wb");
  if (!o) { std::perror("open out"); return 1; }
  std::fwrite(q.data(), 4, q.size(), o); std::fwrite(fk.data(), 4, fk.size(), o);
  std::fwrite(pe.data(), 4, pe.size(), o); std::fwrite(re.data(), 4, re.size(), o);
  std::fwrite(q2.data(), 4, q2.size(), o);
  std::vector<float> vf(n);
  int64_t n_valid = 0;
  for (int64_t i = 0; i < n; ++i) { vf[i] = v2[i] ? 1.f : 0.f; n_valid += v2[i]; }
  std::fwrite(vf.data(), 4, vf.size(), o);
  std::fclose(o);
  if (std::memcmp(q2.data(), q3.data(), q2.size() * 4) != 0 || std::memcmp(v2.data(), v3.data(), v2.size()) != 0 || stats[3] != n_valid) {
    std::fprintf(stderr, "ikf_generate_exact and ikf_refine_exact disagree on identical seeds\n");
    return 4;
  }
  std::printf("cabi_demo: exact IK %lld / %lld poses solved (flow rows %lld)\n", (long long)n_valid, (long long)n, (long long)stats[1]);
  std::printf("cabi_demo: %lld solutions, q[0] = %.6f %.6f %.6f ..., abi %d\n", (long long)n, q[0], q[1], q[2], ikf_abi_version());
  ikf_destroy(m);
  return 0;
}
