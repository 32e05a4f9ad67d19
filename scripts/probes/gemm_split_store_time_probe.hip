// gemm_split_store_time_probe.hip — stand-alone probe (NOT part of the library) of the split-mode NT
// GEMM that converts at LDS-store time (CHANGELOG.md §3 "experiment recorded", §9 item 1):
//   C[M, N] = relu?(A[M, K] @ B[N, K]^T + bias), fp32 in / out, every operand as three bf16 terms,
//   six v_mfma_f32_32x32x16_bf16 products per 16 k values, fp32 accumulation.
// It includes the library's gemm.hip for the shared pieces (argument block, tile numbering, the
// split helpers) and carries the kernel that was taken out of the library at the end of round 2
// (an inline-assembly variant of its loads had faulted once under the profiler and there was no
// GPU time left to re-validate a fix) — here with compiler-managed loads.  The program checks the
// result against a float64 evaluation on the host and times the kernel with its phases switched
// off one by one (argument `--ablate`), which is the experiment §9 item 1 asks for next.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../pytorch_geometric_amd/csrc \
//            -o gemm_split_probe gemm_split_store_time_probe.hip
// Run:   ./gemm_split_probe            (correctness at M = 4133, then timings at M = 2,449,029)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../pytorch_geometric_amd/csrc/gemm.hip"

namespace pygamd {

thread_local int g_last_hip_error = 0;  // (defined in capi.hip in the library)

// ---- NT, split arithmetic, conversion at LDS-store time --------------------------------------------
// The first split kernel above converts a fragment in the wave that multiplies it: every operand
// element is converted twice (two waves share it) and the kernel is VALU-issue bound (9 VALU per
// matrix instruction, matrix pipe 31 % busy).  Here every element is converted ONCE, by the thread
// that stages it, and LDS holds three bf16 planes per operand: a fragment is then a 16-byte
// ds_read per plane, nothing else.  Shape: 512 threads = 8 waves as 4 (M) x 2 (N), workgroup tile
// 256 x 128 (the weight tile serves twice as many rows), wave tile 64 x 64, K in steps of 16 = ONE
// bf16 matrix step: per step and wave 24 matrix instructions, 12 ds_read_b128, and per thread
// 3 global 16-byte loads (two steps ahead), 54 conversion VALU, 9 ds_write_b64.
constexpr int kS3Threads = 512;
constexpr int kS3BM = 256, kS3BN = 128, kS3K = 16;
constexpr int kS3Row = 48;  // bytes per staged row of one plane: 16 bf16 + 16 bytes of padding
                            // (row i starts at bank 12 i: 16 rows x 16 bytes cover all 64 banks)
constexpr int kS3PlaneA = kS3BM * kS3Row, kS3PlaneB = kS3BN * kS3Row;
constexpr int kS3Buf = 3 * (kS3PlaneA + kS3PlaneB);  // bytes per ring slot

// four consecutive k -> 8 bytes in each of the three planes
__device__ __forceinline__ void split4_store(const f32x4& v, char* row_ptr, int plane_stride) {
  uint32_t w[3][2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float x0 = v[2 * q], x1 = v[2 * q + 1];
    const uint32_t a = pack_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(a << 16);
    const float r1 = x1 - __uint_as_float(a & 0xffff0000u);
    const uint32_t b = pack_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(b << 16);
    const float s1 = r1 - __uint_as_float(b & 0xffff0000u);
    w[0][q] = a;
    w[1][q] = b;
    w[2][q] = pack_bf16(s0, s1);
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 o = {w[t][0], w[t][1]};
    *reinterpret_cast<u32x2*>(row_ptr + t * plane_stride) = o;
  }
}

template <bool VEC>
__global__ void __launch_bounds__(kS3Threads, 2) gemm_nt_split3_kernel(GemmNT p) {
  extern __shared__ __align__(16) float smem[];
  char* lds = reinterpret_cast<char*>(smem);  // [2][A planes | B planes]
  int tile_m, tile_n;
  nt_tile_of_block(p, tile_m, tile_n);
  if (tile_m >= p.tiles_m) return;
  const int64_t m0 = static_cast<int64_t>(tile_m) * kS3BM;
  const int n0 = tile_n * kS3BN;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  // staging map: thread -> row t >> 2 (+128 for the second A load), 16-byte column t & 3
  const int sq = threadIdx.x & 3, sr = threadIdx.x >> 2;
  const float* pa[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int64_t row = m0 + sr + 128 * q;
    row = row < p.M ? row : p.M - 1;
    pa[q] = p.a + row * p.lda;
  }
  const bool okb = n0 + sr < p.N;
  const float* pb = p.b + static_cast<int64_t>(okb ? n0 + sr : p.N - 1) * p.ldb;
  struct Stage {
    f32x4 a[2], b;
  };
  auto load_step = [&](Stage& st, int k0) {  // issue only; masked when it moves to LDS
    const int k = k0 + 4 * sq;
    if (VEC) {
      const int kc = k < p.K ? k : 0;
      st.a[0] = *reinterpret_cast<const f32x4*>(pa[0] + kc);
      st.a[1] = *reinterpret_cast<const f32x4*>(pa[1] + kc);
      st.b = *reinterpret_cast<const f32x4*>(pb + kc);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kc = k + e < p.K ? k + e : 0;
        st.a[0][e] = pa[0][kc];
        st.a[1][e] = pa[1][kc];
        st.b[e] = pb[kc];
      }
    }
  };
  auto store_step = [&](const Stage& st, int slot, int k0) {
    const int k = k0 + 4 * sq;
    char* base = lds + slot * kS3Buf;
    f32x4 va[2] = {st.a[0], st.a[1]}, vb = st.b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool kv = k + e < p.K;  // both operands zero past K (Inf * 0 would be NaN)
      va[0][e] = kv ? va[0][e] : 0.f;
      va[1][e] = kv ? va[1][e] : 0.f;
      vb[e] = (kv && okb) ? vb[e] : 0.f;
    }
    split4_store(va[0], base + sr * kS3Row + 8 * sq, kS3PlaneA);
    split4_store(va[1], base + (sr + 128) * kS3Row + 8 * sq, kS3PlaneA);
    split4_store(vb, base + 3 * kS3PlaneA + sr * kS3Row + 8 * sq, kS3PlaneB);
  };
  struct Frags {
    SplitFrag a[2], b[2];
  };
  auto read_frags = [&](Frags& f, int slot) {
    const char* base = lds + slot * kS3Buf;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        f.a[i].p[t] = *reinterpret_cast<const bf16x8*>(base + t * kS3PlaneA +
                                                       (wm * 64 + i * 32 + li) * kS3Row + 16 * lh);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        f.b[j].p[t] = *reinterpret_cast<const bf16x8*>(base + 3 * kS3PlaneA + t * kS3PlaneB +
                                                       (wn * 64 + j * 32 + li) * kS3Row + 16 * lh);
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ring of two LDS slots, global loads two steps ahead of their store:
  //   iteration c: fragments <- LDS[c];  LDS[c+1] <- convert(s0);  s0 <- s1;  s1 <- global step c+3;
  //                24 matrix instructions on the fragments;  barrier.
  const int n_steps = (p.K + kS3K - 1) / kS3K;
  // Pipeline (measured floor of a version that read its fragments at the top of every step: the
  // 96 KB of ds_read_b128 per step and CU cost as much as half of the step's matrix time, exposed).
  //   iteration c:  fragments(c + 1) <- LDS[(c + 1) & 1]      (issued first, land under the MFMAs)
  //                 step c + 2: wait for its loads (issued at iteration c - 2), convert, store
  //                             into LDS[c & 1] (whose fragments were read during iteration c - 1)
  //                 loads of step c + 4 -> the staging registers just freed
  //                 24 matrix instructions on fragments(c), the 18 conversion slices between them
  //                 barrier (step c + 2 visible; everybody is done reading LDS[(c + 1) & 1])
  // Registers alternate by parity — staging sE / sO (a step waits in them for two iterations, no
  // copies: a copy would wait for the youngest load at once) and fragments fE / fO — so the loop
  // is unrolled by two.  Loads past the last step are clamped to valid addresses and their stores
  // write zeros nobody reads: every iteration is the same straight-line block and the explicit
  // vmcnt(3) (three younger loads in flight) is exact.
  Stage sE, sO;
  load_step(sE, 0);
  load_step(sO, kS3K);
  store_step(sE, 0, 0);
  store_step(sO, 1, kS3K);
  load_step(sE, 2 * kS3K);
  load_step(sO, 3 * kS3K);
  __syncthreads();
  Frags fE, fO;
  read_frags(fE, 0);
  char* const st_a0 = lds + sr * kS3Row + 8 * sq;
  char* const st_a1 = lds + (sr + 128) * kS3Row + 8 * sq;
  char* const st_b = lds + 3 * kS3PlaneA + sr * kS3Row + 8 * sq;
  // st: staging registers of step c + 2 (then of c + 4);  f: fragments(c);  fn: fragments(c + 1)
  auto iteration = [&](int c, Stage& st, const Frags& f, Frags& fn) {
    const int slot = c & 1;
    __builtin_amdgcn_sched_barrier(0);
    read_frags(fn, slot ^ 1);
    // pairs 0,1 = first A row, 2,3 = second A row, 4,5 = B row of the step being staged
    float xi[6][2], rr[6][2], ss[6][2];
    uint32_t w0[6], w1[6], w2[6];
    {
      const int k = (c + 2) * kS3K + 4 * sq;
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const f32x4& src = q < 2 ? st.a[0] : (q < 4 ? st.a[1] : st.b);
          const bool kv = k + 2 * (q & 1) + e < p.K && (q < 4 || okb);
          xi[q][e] = kv ? src[2 * (q & 1) + e] : 0.f;
        }
    }
    if (!(p.split & 64)) load_step(st, (c + 4) * kS3K);
    auto slice = [&](int n) {  // n = 6 * level + pair
      const int q = n % 6;
      if (p.split & 32) return;
      if (n < 6) {
        w0[q] = pack_bf16(xi[q][0], xi[q][1]);
        rr[q][0] = xi[q][0] - __uint_as_float(w0[q] << 16);
        rr[q][1] = xi[q][1] - __uint_as_float(w0[q] & 0xffff0000u);
      } else if (n < 12) {
        w1[q] = pack_bf16(rr[q][0], rr[q][1]);
        ss[q][0] = rr[q][0] - __uint_as_float(w1[q] << 16);
        ss[q][1] = rr[q][1] - __uint_as_float(w1[q] & 0xffff0000u);
      } else {
        w2[q] = pack_bf16(ss[q][0], ss[q][1]);
        if (q & 1) {  // both pairs of a 16-byte load are done: 8 bytes into each plane
          typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
          char* dst = (q == 1 ? st_a0 : (q == 3 ? st_a1 : st_b)) + slot * kS3Buf;
          const int ps = q == 5 ? kS3PlaneB : kS3PlaneA;
          *reinterpret_cast<u32x2*>(dst) = u32x2{w0[q - 1], w0[q]};
          *reinterpret_cast<u32x2*>(dst + ps) = u32x2{w1[q - 1], w1[q]};
          *reinterpret_cast<u32x2*>(dst + 2 * ps) = u32x2{w2[q - 1], w2[q]};
        }
      }
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      const int t = m >> 2, i = (m >> 1) & 1, j = m & 1;
      if (!(p.split & 16)) split_term(t, f.a[i], f.b[j], acc[i][j]);
      if (m < 18) slice(m);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
  int c = 0;
  for (; c + 1 < n_steps; c += 2) {
    iteration(c, sE, fE, fO);
    iteration(c + 1, sO, fO, fE);
  }
  if (c < n_steps) iteration(c, sE, fE, fO);
  // plain epilogue of the probe: bias + optional ReLU, bounds-checked
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + (wn * 2 + j) * 32 + li;
      if (col >= p.N) continue;
      const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = m0 + (wm * 2 + i) * 32 + 4 * lh + (e & 3) + 8 * (e >> 2);
        if (row < p.M) {
          const float v = acc[i][j][e] + bv;
          p.c[row * p.ldc + col] = p.relu ? fmaxf(v, 0.f) : v;
        }
      }
    }
}


}  // namespace pygamd

using namespace pygamd;

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

static float run(const float* dA, const float* dB, const float* dbias, float* dC, int64_t M, int N,
                 int K, int relu, int ablate, int reps) {
  GemmNT p = {};
  p.a = dA; p.b = dB; p.bias = dbias; p.c = dC;
  p.M = M; p.lda = K; p.ldb = K; p.ldc = N; p.N = N; p.K = K; p.relu = relu;
  p.split = 1 | ablate;
  p.tiles_m = static_cast<int>((M + kS3BM - 1) / kS3BM);
  p.tiles_n = (N + kS3BN - 1) / kS3BN;
  const int64_t blocks = (static_cast<int64_t>(p.tiles_m) * p.tiles_n + 7) / 8 * 8;
  const size_t lds = 2 * kS3Buf;
  auto k = gemm_nt_split3_kernel<true>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(static_cast<unsigned>(blocks)), dim3(kS3Threads), lds, 0, p);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(k, dim3(static_cast<unsigned>(blocks)), dim3(kS3Threads), lds, 0, p);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char** argv) {
  const bool ablate = argc > 1 && std::string(argv[1]) == "--ablate";
  // ---- correctness: ragged M (partial last tile), K with a 16-tail, N = 256 and 200 ----
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int64_t M = 4133;
    const int N = cfg == 2 ? 200 : 256, K = cfg == 0 ? 512 : 200;
    std::mt19937_64 rng(7 + cfg);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> A(M * K), B(static_cast<size_t>(N) * K), bias(N);
    for (auto& v : A) v = nd(rng);
    for (auto& v : B) v = nd(rng) * 0.1f;
    for (auto& v : bias) v = nd(rng);
    float *dA, *dB, *dbias, *dC;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4));
    CK(hipMalloc(&dbias, N * 4)); CK(hipMalloc(&dC, M * N * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, bias.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0xff, M * N * 4));
    run(dA, dB, dbias, dC, M, N, K, 1, 0, 1);
    std::vector<float> C(M * N);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int64_t i = 0; i < M; i += 7)
      for (int j = 0; j < N; ++j) {
        double s = bias[j], a = std::fabs(bias[j]);
        for (int k = 0; k < K; ++k) {
          const double t = static_cast<double>(A[i * K + k]) * B[static_cast<size_t>(j) * K + k];
          s += t; a += std::fabs(t);
        }
        s = s > 0 ? s : 0;
        worst = std::fmax(worst, std::fabs(C[i * N + j] - s) / a);
      }
    printf("check M=%lld N=%d K=%d: max |err| / sum|ab| = %.3e  %s\n", static_cast<long long>(M), N,
           K, worst, worst < 2e-6 ? "OK" : "FAILED");
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dbias)); CK(hipFree(dC));
  }
  // ---- timing at the products shape ----
  const int64_t M = 2449029;
  for (int K : {512, 200}) {
    const int N = 256;
    float *dA, *dB, *dbias, *dC;
    CK(hipMalloc(&dA, M * K * 4)); CK(hipMalloc(&dB, static_cast<size_t>(N) * K * 4));
    CK(hipMalloc(&dbias, N * 4)); CK(hipMalloc(&dC, M * N * 4));
    CK(hipMemset(dA, 0x3c, M * K * 4));  // finite fp32 pattern
    CK(hipMemset(dB, 0x3c, static_cast<size_t>(N) * K * 4));
    CK(hipMemset(dbias, 0, N * 4));
    const double gf = 2.0 * M * K * N * 1e-9;
    const int modes[] = {0, 16, 32, 64, 48, 112};
    const char* names[] = {"full kernel", "no matrix instructions", "no conversion / LDS stores",
                           "no global loads", "no matrix instr., no conversion",
                           "skeleton (fragment reads + barriers + epilogue)"};
    for (int m = 0; m < (ablate ? 6 : 1); ++m) {
      const float ms = run(dA, dB, dbias, dC, M, N, K, 1, modes[m], 5);
      printf("K=%3d  %-48s %7.3f ms  %6.1f TFLOP/s fp32-equivalent\n", K, names[m], ms, gf / ms);
    }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dbias)); CK(hipFree(dC));
  }
  return 0;
}
