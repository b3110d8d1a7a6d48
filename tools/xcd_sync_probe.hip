// What does an in-launch hand-over cost when the workgroups that exchange data share an XCD (and therefore an L2)?
// 256 persistent workgroups (one per CU).  Each reads its XCC_ID, takes a ticket from its XCD's counter (so the grouping is whatever the
// dispatcher did - nothing is assumed about workgroup -> XCD placement), then runs N rounds of
//     write 256 B  ->  drain  ->  arrive on the group's counter  ->  spin until the whole group arrived  ->  read every member's 256 B, verify
// in three flavours:
//   mode 0  groups = XCDs; plain stores (they sit in the XCD's L2), agent-scope relaxed atomics / loads (L1 miss, L2 hit), NO fences
//   mode 1  one group = the whole chip (256 workgroups); write-through stores (sc1) + agent-scope loads - the TailSync protocol of the library
//   mode 2  groups = XCDs, but with the chip-wide protocol's stores (sc1 write-through): isolates the store flavour
//   mode 3  groups = XCDs; plain stores; after the spin ONE agent-scope acquire fence (buffer_inv sc1: drops the CU's L1), then PLAIN loads -
//           the form that lets unmodified kernel bodies (ordinary cached loads) run behind the hand-over
// Prints us per round (shader clock) and the number of stale / wrong values read (must be 0 for a flavour to be usable).
// Spins are bounded: a workgroup that gives up sets a flag and everyone drains.
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_sync_probe.hip -o tools/bin/xcd_sync_probe && timeout 60 tools/bin/xcd_sync_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Ctl {
  unsigned ticket[8 * 32];   // [xcd][0]: one cache line (128 B) per XCD
  unsigned arrive[9 * 32];   // [group][0]; group 8 = whole chip
  unsigned give_up;
};

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf; }  // HW_REG_XCC_ID

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(Ctl* c, unsigned* data /* [9][2][256][64] */, int rounds, unsigned long long* out /* [256][4] */) {
  __shared__ unsigned s_ticket, s_ok;
  const int t = threadIdx.x;
  const unsigned xcd = xcc_id();
  if (t == 0) s_ticket = __hip_atomic_fetch_add(&c->ticket[xcd * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const unsigned ticket = s_ticket;
  const unsigned group = MODE == 1 ? 8 : xcd;
  // members of the group: mode 1 needs a chip-wide index: (xcd, ticket) -> xcd * 32 + ticket (valid when every XCD got 32)
  const unsigned me = MODE == 1 ? xcd * 32 + ticket : ticket;
  const unsigned members = MODE == 1 ? 256 : 32;
  unsigned* gdata = data + (size_t)group * 2 * 256 * 64;
  unsigned long long bad = 0, spins = 0;
  const unsigned long long t0 = wall_clock64();
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int r = 1; r <= rounds; ++r) {
    unsigned* slot = gdata + (size_t)(r & 1) * 256 * 64;
    if (t < 64) {
      const unsigned v = (unsigned)r * 1024u + me;
      if (MODE == 0 || MODE == 3) slot[me * 64 + t] = v;
      else __hip_atomic_store(&slot[me * 64 + t], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1: written through
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // s_waitcnt vmcnt(0): the stores are acknowledged by the L2
    __syncthreads();
    if (t == 0) {
      __hip_atomic_fetch_add(&c->arrive[group * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned ok = 1, n = 0;
      while (__hip_atomic_load(&c->arrive[group * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < members * (unsigned)r) {
        if (++n > (1u << 18) || __hip_atomic_load(&c->give_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
      }
      spins += n;
      s_ok = ok;
      if (!ok) __hip_atomic_store(&c->give_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_ok) break;
    // read every member's block: thread t reads word t % 64 of members t / 64, t / 64 + 4, ...
    if (MODE == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (unsigned m = t >> 6; m < members; m += 4) {
      const unsigned v = MODE == 3 ? *(volatile unsigned*)&slot[m * 64 + (t & 63)]
                                   : __hip_atomic_load(&slot[m * 64 + (t & 63)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != (unsigned)r * 1024u + m) ++bad;
    }
  }
  const unsigned long long t1 = wall_clock64();
  const unsigned long long c1 = __builtin_readcyclecounter();
  // per-thread bad counts -> one per workgroup
  __shared__ unsigned long long s_bad;
  if (t == 0) s_bad = 0;
  __syncthreads();
  if (bad) atomicAdd(&s_bad, bad);
  __syncthreads();
  if (t == 0) {
    out[blockIdx.x * 6 + 0] = xcd;
    out[blockIdx.x * 6 + 1] = ticket;
    out[blockIdx.x * 6 + 2] = t1 - t0;
    out[blockIdx.x * 6 + 3] = s_bad;
    out[blockIdx.x * 6 + 4] = spins;
    out[blockIdx.x * 6 + 5] = c1 - c0;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
  Ctl* c; unsigned* data; unsigned long long* out;
  CK(hipMalloc(&c, sizeof(Ctl)));
  CK(hipMalloc(&data, sizeof(unsigned) * 9 * 2 * 256 * 64));
  CK(hipMalloc(&out, sizeof(unsigned long long) * 256 * 6));
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemset(c, 0, sizeof(Ctl)));
      CK(hipMemset(data, 0, sizeof(unsigned) * 9 * 2 * 256 * 64));
      if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(256), dim3(256), 0, 0, c, data, rounds, out);
      if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(256), dim3(256), 0, 0, c, data, rounds, out);
      if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3(256), dim3(256), 0, 0, c, data, rounds, out);
      if (mode == 3) hipLaunchKernelGGL(k_probe<3>, dim3(256), dim3(256), 0, 0, c, data, rounds, out);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> h(256 * 6);
      Ctl hc;
      CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(&hc, c, sizeof(Ctl), hipMemcpyDeviceToHost));
      int per_xcd[16] = {};
      unsigned long long bad = 0, wall = 0, spins = 0, cyc = 0;
      for (int b = 0; b < 256; ++b) {
        per_xcd[h[b * 6] & 15]++;
        bad += h[b * 6 + 3];
        if (h[b * 6 + 2] > wall) wall = h[b * 6 + 2];
        if (h[b * 6 + 5] > cyc) cyc = h[b * 6 + 5];
        spins += h[b * 6 + 4];
      }
      printf("mode %d rep %d: %.3f us per round (wall clock, 100 MHz), %.0f shader cycles per round; stale/wrong values %llu; polls per round per workgroup %.1f; give_up %u; workgroups per XCD:",
             mode, rep, wall / 100.0 / rounds, (double)cyc / rounds, bad, (double)spins / rounds / 256, hc.give_up);
      for (int x = 0; x < 8; ++x) printf(" %d", per_xcd[x]);
      printf("\n");
      if (rep == 0) {
        printf("   first workgroups (block -> xcd:ticket):");
        for (int b = 0; b < 24; ++b) printf(" %d->%llu:%llu", b, h[b * 6], h[b * 6 + 1]);
        printf("\n");
      }
    }
  }
  return 0;
}
