// kernels_fast.hip -- fast-mode HIP kernels for gfx950: the throughput path.
//
// Same model as the reference (two-state HMM with distance-dependent transitions,
// shared/HMM.cpp), reformulated for the hardware:
//
//  * LINEAR space with rescaling instead of log space.  The reference spends 14
//    exp/log calls per site per recursion (HMM.cpp:130-139, gen_func.cpp:135-151);
//    here a site costs one exp (e^{-alpha d}) and ~8 FMAs per tracked vector.
//    Per-call results agree with the log-space oracle to ~1e-13 relative (the
//    tests allow 1e-9); the running magnitude is kept in an integer exponent.
//
//  * CHUNK-PARALLEL over sites.  A 2-state forward step is a 2x2 linear operator
//      M_s = (c_s I + (1-c_s) 1 q^T) diag(e_s),   c_s = exp(-alpha d_s),
//    so a run of sites is the product of its operators.  An individual's sites are
//    cut into J = 64*C runs of T sites; LANE j of the individual's C waves walks
//    run j sequentially and the 64*C per-lane operators are combined afterwards
//    (in-wave ordered shuffle tree, then a tiny cross-wave pass).  No lane ever
//    waits for another inside the main loop.
//
//  * INTERLEAVED layout [I][C][T][64]: element (t, lane) of wave (i, c) is site
//    (c*64 + lane)*T + t.  Every load of the main loops is one contiguous
//    1 KiB (double2 emissions) or 512 B segment per wave-instruction.
//
//  * MATERIALISED emission RATIOS (8 B per site-individual) rather than recomputing the
//    emissions from the 24 B genotype likelihoods in every pass of an iteration.  Scaling both
//    emissions of a site by a common factor changes neither posteriors nor paths and adds
//    the factor's log to the likelihood, whatever (indF, alpha) are: with
//      rho_s = e1_s / e0_s,   base = sum_s log e0_s
//    every walk runs on the emissions (1, rho_s) and the likelihood is base + the walk's
//    result.  rho and base are refreshed by the first forward walk after a frequency
//    update, on its way; base is one number per wave (individual, chunk).
//
//  * the <= 5 probe points of one individual's finite-difference gradient
//    (shared/bfgs.cpp:22-43) share ONE pass over that individual's emissions, in a
//    kernel specialised for the pattern of the points.
//
//  * ONE forward walk per EM iteration serves the M-step's first objective round, the
//    E-step (it leaves lane operators and a checkpoint every 8 sites; the backward
//    sweep recomputes forward vectors block-wise) and the emission refresh.
//
//  * posteriors in a TILE-MAJOR layout ([tile row c*T + t][i / 8][l][i % 8]: kPost8 below) the
//    backward sweep writes with contiguous stores; est_maf reads it in place.
//
// The streaming kernels are HBM-bound, the objective rounds and est_maf FP64-issue
// bound (no MFMA: there is no contraction longer than 2).  DESIGN.md section 4.
#include <hip/hip_runtime.h>

#include "fast_dev.hpp"

namespace nghmm {

namespace {


// site-major codes [S][I] (2 bits per cell) -> interleaved words [I][C][T/16][64]: word
// (i, c, tb, lane) holds the codes of sites (c*64 + lane)*T + tb*16 .. + 15.  Padding sites
// get code 0, which the padding frequency 0 turns into the identity emission (1, 1).
__global__ void __launch_bounds__(256)
k_fast_geno_interleave(const uint32_t* __restrict__ codes, uint64_t cell0, uint64_t I, uint64_t S,
                       uint64_t T, uint32_t C, uint32_t* __restrict__ geno_il) {
  const uint64_t TB = T >> 4;
  const uint64_t n = I * C * TB * 64;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t lane = k & 63, r = k >> 6;
    const uint64_t tb = r % TB, wv = r / TB;
    const uint64_t c = wv % C, i = wv / C;
    const uint64_t s0 = (c * 64 + lane) * T + tb * 16;
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint64_t s = s0 + j;
      if (s < S) w |= gl_code(codes, cell0 + s * I + i) << (2 * j);
    }
    geno_il[k] = w;
  }
}

// linear-space copy of the genotype likelihoods: they do not change during a run, and
// both the emission refresh and est_maf would otherwise exponentiate all 3 I S of them
// in every EM iteration
__global__ void __launch_bounds__(256)
k_fast_exp(const double* in, double* out, uint64_t n) {  // in == out allowed
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (uint64_t)gridDim.x * blockDim.x)
    out[k] = exp(in[k]);
}

// largest finite distance (bit pattern order == value order for non-negative doubles)
__global__ void __launch_bounds__(256)
k_fast_max_finite(const double* __restrict__ pos, uint64_t S, unsigned long long* __restrict__ out) {
  unsigned long long m = 0;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < S;
       k += (uint64_t)gridDim.x * blockDim.x) {
    const double d = pos[k];
    if (d >= 0 && d < kDStart) {
      const unsigned long long b = ngh_bits(d);
      m = b > m ? b : m;
    }
  }
  atomicMax(out, m);
}

__global__ void __launch_bounds__(256)
k_fast_pos_interleave(const double* __restrict__ pos, uint64_t S, uint64_t T, uint32_t C,
                      double* __restrict__ pos_il) {
  const uint64_t n = (uint64_t)C * T * 64;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t lane = k & 63, ct = k >> 6;
    const uint64_t c = ct / T, t = ct % T;
    const uint64_t s = (c * 64 + lane) * T + t;
    // padding: d = 0 -> c = 1 -> identity transition; chromosome starts: +inf is stored
    // as kDStart = 1e22 (exp(-alpha 1e22) = 0 for every alpha >= 1e-15, and no inf*0 can arise)
    const double d = (s < S) ? pos[s] : 0.0;
    pos_il[k] = (d < kDStart) ? d : kDStart;
  }
}

// What the kappa-form walks (fast_dev.hpp: op_step_k) leave out of a lane-chunk's operator: the
// sum of its finite distances, in site order, and the number of its chromosome starts.
__global__ void __launch_bounds__(64)
k_fast_chunk_scale(const double* __restrict__ pos_il, uint64_t T, double2* __restrict__ chunk_scale) {
  const double* dp = pos_il + (uint64_t)blockIdx.x * T * 64 + threadIdx.x;
  double sum = 0.0, starts = 0.0;
  for (uint64_t t = 0; t < T; ++t) {
    const double d = dp[t * 64];
    if (d < kDStart) sum += d;
    else starts += 1.0;
  }
  chunk_scale[(uint64_t)blockIdx.x * 64 + threadIdx.x] = double2{sum, starts};
}

// freq[S] -> interleaved [C][T][64] for the fresh forward walk (padding: 0, which makes both
// the dense padding likelihoods (1, 1, 1) and the packed padding code 0 the identity emission
// (1, 1) exactly); flags an
// allele frequency outside [0, 1] like calc_emission does (shared/HMM.cpp:145-146)
__global__ void __launch_bounds__(256)
k_fast_freq_interleave(const double* __restrict__ freq, uint64_t S, uint64_t T, uint32_t C,
                       double* __restrict__ freq_il, int* __restrict__ flags) {
  const uint64_t n = (uint64_t)C * T * 64;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t lane = k & 63, ct = k >> 6;
    const uint64_t c = ct / T, t = ct % T;
    const uint64_t s = (c * 64 + lane) * T + t;
    double f = 0.0;
    if (s < S) {
      f = freq[s];
      if (f < 0 || f > 1) {  // the reference's predicate (HMM.cpp:145): NaN passes, and
                             // surfaces later as "invalid Lkl found!" like in exact mode
        flags[FLAG_INVALID_MAF] = 1;
        f = __builtin_nan("");
      }
    }
    freq_il[k] = f;
  }
}

// site-major LOG GL [S][I][3] -> interleaved, each cell relative to its largest likelihood
// (glq_encode), layout of e_il; padding sites get (1, 1, 1), which any frequency turns into
// the identity emission (1, 1).
// tile = (c, t) x 64 lanes (sites T apart) x 32 individuals
__global__ void __launch_bounds__(256)
k_fast_gl_interleave(const double* __restrict__ gl_log, uint64_t I, uint64_t S, uint64_t T,
                     uint32_t C, double2* __restrict__ glq_il) {
  __shared__ double2 tq[32][65];
  const uint64_t n_it = (I + 31) / 32;
  const uint64_t ct = blockIdx.x / n_it;
  const uint64_t i0 = (blockIdx.x % n_it) * 32;
  const uint64_t c = ct / T, t = ct % T;
  {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 individuals x 8
    const uint64_t i = i0 + tx;
    for (int ll = ty; ll < 64; ll += 8) {
      const uint64_t s = (c * 64 + ll) * T + t;
      double2 q{1.0, 1.0};
      if (s < S && i < I) {
        const double* g = gl_log + (s * I + i) * 3;
        q = glq_encode(g[0], g[1], g[2]);
      }
      tq[tx][ll] = q;
    }
  }
  __syncthreads();
  {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 lanes x 4
    for (int ii = ty; ii < 32; ii += 4) {
      const uint64_t i = i0 + ii;
      if (i < I) glq_il[((i * C + c) * T + t) * 64 + tx] = tq[ii][tx];
    }
  }
}

// scale_c[i][c] = sum over the sites of wave (i, c) of the cell's largest log likelihood (what
// glq_encode divides out).  Block = chunk c x 32 individuals; a thread adds its sites in
// order, the eight partial sums of an individual are added in order: the same bits every run.
__global__ void __launch_bounds__(256)
k_fast_gl_scale(const GlView gl_log, uint64_t I, uint64_t S, uint64_t T, uint32_t C,
                double* __restrict__ scale_c) {
  __shared__ double part[8][33];
  const uint64_t n_it = (I + 31) / 32;
  const uint64_t c = blockIdx.x / n_it;
  const uint64_t i0 = (blockIdx.x % n_it) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const uint64_t i = i0 + tx;
  double acc = 0;
  if (i < I) {
    for (uint64_t j = c * 64 + ty; j < c * 64 + 64; j += 8) {
      for (uint64_t t = 0; t < T; ++t) {
        const uint64_t s = j * T + t;
        if (s >= S) break;
        double g0, g1, g2;
        gl_fetch(gl_log, s * I + i, g0, g1, g2);
        acc += fmax(g0, fmax(g1, g2));
      }
    }
  }
  part[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && i < I) {
    double sum = 0;
    for (int k = 0; k < 8; ++k) sum += part[k][tx];
    scale_c[i * C + c] = sum;
  }
}

// out[i][s][k] = log emission of state k (test/debug read-back): recomputed from the
// likelihoods (or interleaved codes) and frequencies with the fresh walk's expressions --
// only the ratio of the two is kept between walks
__global__ void __launch_bounds__(256)
k_fast_export_e(LklArrays arr, bool packed, const double* __restrict__ gl_lin, uint64_t I,
                uint64_t S, uint64_t T, uint32_t C, double* __restrict__ out) {
  const uint64_t n = I * S;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t i = k / S, s = k % S;
    const uint64_t j = s / T, t = s % T;
    const uint64_t c = j >> 6, lane = j & 63;
    const uint64_t wv = i * C + c;
    const double maf = arr.freq_il[(c * T + t) * 64 + lane], om = 1 - maf;
    const double bb = om * maf;
    const double h00 = om * om, h02 = maf * maf;
    double e0, e1;
    if (packed) {
      const uint32_t w = arr.geno_il[(wv * (T >> 4) + (t >> 4)) * 64 + lane];
      const uint32_t code = (w >> ((uint32_t)(t & 15) * 2)) & 3u;
      const double u = arr.u_lin;
      const double u0 = fma(u, h00, fma(u, 2 * bb, u * h02));
      const double u1 = fma(u, h00 + bb, u * (h02 + bb));
      e0 = code == 0 ? h00 : code == 1 ? 2 * bb : code == 2 ? h02 : u0;
      e1 = code == 0 ? h00 + bb : code == 1 ? 0.0 : code == 2 ? h02 + bb : u1;
    } else {
      const double* g = gl_lin + (s * I + i) * 3;  // the site-major linear copy est_maf reads
      e0 = fma(g[0], h00, fma(g[1], 2 * bb, g[2] * h02));
      e1 = fma(g[0], h00 + bb, g[2] * (h02 + bb));
    }
    out[k * 2] = log(e0);
    out[k * 2 + 1] = log(e1);
  }
}

struct SwitchName {
  const char* name;
  int Switches::*field;
};
const SwitchName kSwitches[] = {
    {"bg_parts", &Switches::bg_parts}, {"estmaf_interp", &Switches::estmaf_interp},
    {"estmaf_no_rows", &Switches::estmaf_no_rows}, {"estmaf_no_called", &Switches::estmaf_no_called},
    {"fast_c", &Switches::fast_c}, {"exact_serial", &Switches::exact_serial},
    {"dbg_abort_round", &Switches::dbg_abort_round}, {"no_dev_bfgs", &Switches::no_dev_bfgs},
    {"no_bg_stream", &Switches::no_bg_stream}, {"spans", &Switches::spans}};

}  // namespace

// ---------------------------------------------------------------------------
Switches Switches::from_env() {
  Switches sw;
  for (const auto& k : kSwitches) {
    std::string env = "NGHMM_";
    for (const char* c = k.name; *c; ++c) env += (char)std::toupper((unsigned char)*c);
    if (const char* v = std::getenv(env.c_str())) {
      char* end = nullptr;
      const long n = std::strtol(v, &end, 10);
      sw.*(k.field) = (end != v) ? (int)n : 1;  // set without a number: on
    }
  }
  return sw;
}

bool Switches::set(const char* name, long value) {
  for (const auto& k : kSwitches)
    if (std::strcmp(k.name, name) == 0) {
      this->*(k.field) = (int)value;
      return true;
    }
  return false;
}

// the arrays an EM run writes (everything but the data): emissions, frequency table,
// posteriors, checkpoints, boundary operators
static bool fast_alloc_run_state(FastState& fs) {
  const size_t cells = (size_t)fs.I * fs.Spad;
  const size_t slack = 8 * 64;  // the pipelines read one group past the last lane-chunk
  if (!dalloc(&fs.e_il, cells + slack)) return false;
  if (hipMemset(fs.e_il + cells, 0, slack * sizeof(double)) != hipSuccess) return false;
  if (!dalloc(&fs.base_c, (size_t)fs.I * fs.C)) return false;
  if (!dalloc(&fs.freq_il, (size_t)fs.Spad + slack)) return false;
  if (hipMemset(fs.freq_il + fs.Spad, 0, slack * sizeof(double)) != hipSuccess) return false;
  const size_t post_cells = (size_t)fs.C * fs.T * post_tile_doubles(fs.I);
  if (!dalloc(&fs.post, post_cells)) return false;
  // the reference starts from marg_prob = 0 (parse_args.cpp:403-405): `--freq e` and a
  // `--log` print before the first E-step read the posteriors
  if (hipMemset(fs.post, 0, post_cells * sizeof(double)) != hipSuccess) return false;
  if (!dalloc(&fs.ckpt, cells / CK * 4 + 4 * 64)) return false;  // T is a multiple of CK; slack:
                                                                 // see lkl_run_fd
  if (!dalloc(&fs.lane_ops, (size_t)fs.I * fs.J * 5)) return false;
  if (!dalloc(&fs.bound, (size_t)fs.I * fs.J * 4)) return false;
  fs.e_stale = true;
  return true;
}

bool fast_create(FastState& fs, uint64_t I, uint64_t S, bool packed) {
  fs.I = I;
  fs.S = S;
  fs.packed = packed;
  fs.owns_data = true;
  // Waves per individual.  The objective rounds run 4 waves per SIMD = 4096 at a time and
  // all their waves take equally long, so a launch of n waves wastes the unfilled part of
  // its last batch, and the later rounds launch only the still-active individuals: ~49k waves
  // per 1000 individuals (measured at 1000 x 1M, ms per EM iteration: C = 9: 45, 33: 28.85,
  // 40: 28.3, 48: 28.1-28.3, 56: 28.2, 64: 28.9, 96: 29.0, 128: 29.8 -- beyond ~56 a wave's
  // fixed cost, the operator tree of every point, outweighs the fuller batches).  At least 16
  // sites per lane.
  uint64_t C = (49152 + I - 1) / I;
  if (fs.sw.fast_c >= 1) C = (uint64_t)fs.sw.fast_c;  // tuning knob: waves per individual
  if (C > 256) C = 256;  // (measured at 125 x 1M: 5.50 ms per iteration with 64, 5.30 with 256)
  // at least 32 sites per lane above 64 waves per individual (16 below): a wave's fixed cost
  // -- the 64-lane operator tree of every point -- is that of about a dozen sites
  while (C > 1 && (S + 64 * C - 1) / (64 * C) < (C > 64 ? 32u : 16u)) --C;
  if (C < 1) C = 1;
  // ... and 64 sites per lane where the cohort still fills the chip one and a half times over at
  // that (>= 24576 waves): the site shard of an eight-GPU run, 1000 x 125 000 -- 3.76 ms per
  // iteration at 49 waves per individual (40 sites per lane), 3.63 at 25 (80): with the kappa
  // form a walk ends in three exponentials per lane on top of the operator trees
  if (fs.sw.fast_c < 1) {
    const uint64_t c64 = S / (64 * 64);
    if (c64 >= 1 && c64 < C && I * c64 >= 24576) C = c64;
  }
  if (fs.sw.fast_c < 1) {
    // sites per lane are rounded up to whole groups of 8 (16: packed), which at a few dozen
    // sites per lane pads a lot (100 x 100k: 24.4 -> 32 sites per lane, 31 %): take the count
    // within a quarter below the target that pads least
    const uint64_t q = packed ? 16 : 8;
    auto padded = [&](uint64_t c) { return 64 * c * ((((S + 64 * c - 1) / (64 * c)) + q - 1) / q * q); };
    uint64_t best = C;
    for (uint64_t c = C; c >= 1 && 4 * c >= 3 * C; --c)
      if (padded(c) < padded(best)) best = c;
    C = best;
  }
  fs.C = (uint32_t)C;
  fs.J = 64 * C;
  fs.T = (S + fs.J - 1) / fs.J;
  fs.T = (fs.T + 7) & ~7ull;  // whole prefetch groups: the main loops need no bound checks
  if (packed) fs.T = (fs.T + 15) & ~15ull;  // 16 sites of a lane per code word
  fs.Spad = fs.J * fs.T;
  const size_t cells = (size_t)I * fs.Spad;
  const size_t slack = 8 * 64;  // the pipelines read one group past the last lane-chunk
  if (!dalloc(&fs.pos_il, (size_t)fs.Spad + slack)) return false;
  if (hipMemset(fs.pos_il + fs.Spad, 0, slack * sizeof(double)) != hipSuccess) return false;
  if (!dalloc(&fs.chunk_scale, (size_t)2 * fs.J)) return false;
  if (!dalloc(&fs.gl_scale_c, (size_t)I * C)) return false;
  if (packed) {
    const size_t words = cells / 16 + slack;
    if (!dalloc(&fs.geno_il, words)) return false;
    if (hipMemset(fs.geno_il, 0, words * sizeof(uint32_t)) != hipSuccess) return false;
    if (!dalloc(&fs.cls_lin, (size_t)12)) return false;
  } else {
    if (!dalloc(&fs.glq_il, (cells + slack) * 2)) return false;
    if (hipMemset(fs.glq_il + cells * 2, 0, slack * 2 * sizeof(double)) != hipSuccess) return false;
    if (!dalloc(&fs.gl_lin, (size_t)I * S * 3)) return false;
  }
  return fast_alloc_run_state(fs);
}

bool fast_create_replica(FastState& fs, const FastState& parent) {
  fs = FastState();
  fs.sw = parent.sw;
  fs.I = parent.I;
  fs.S = parent.S;
  fs.T = parent.T;
  fs.C = parent.C;
  fs.J = parent.J;
  fs.Spad = parent.Spad;
  fs.packed = parent.packed;
  fs.owns_data = false;
  fs.gl_log = parent.gl_log;
  fs.d_pos = parent.d_pos;
  fs.geno_il = parent.geno_il;
  fs.cls_lin = parent.cls_lin;
  fs.u_lin = parent.u_lin;
  fs.called_table = parent.called_table;
  fs.gl_lin = parent.gl_lin;
  fs.pos_il = parent.pos_il;
  fs.chunk_scale = parent.chunk_scale;
  fs.glq_il = parent.glq_il;
  fs.gl_scale_c = parent.gl_scale_c;
  fs.dmax_finite = parent.dmax_finite;
  fs.alpha_small_min = parent.alpha_small_min;
  return fast_alloc_run_state(fs);
}

void fast_destroy(FastState& fs) {
  dbfgs_destroy(fs);
  void* run[] = {fs.e_il, fs.base_c, fs.freq_il, fs.post, fs.ckpt, fs.lane_ops, fs.bound, fs.lanes[0].part,
                 fs.lanes[0].grp_dev, fs.lanes[1].part, fs.lanes[1].grp_dev, fs.redo, fs.est_status,
                 fs.est_state, fs.est_counts, fs.shard.edges};
  for (void* p : run)
    if (p) (void)hipFree(p);
  if (fs.owns_data) {
    void* data[] = {fs.pos_il, fs.chunk_scale, fs.glq_il, fs.gl_scale_c, fs.gl_lin, fs.geno_il, fs.cls_lin};
    for (void* p : data)
      if (p) (void)hipFree(p);
  }
  fs = FastState();
}

void fast_exp(hipStream_t st, const double* d_in, double* d_out, uint64_t n) {
  if (n) hipLaunchKernelGGL(k_fast_exp, dim3(16384), dim3(256), 0, st, d_in, d_out, n);
}

GlView fast_gl_lin(const FastState& fs) {
  return fs.packed ? gl_packed(fs.gl_log.codes, fs.cls_lin, fs.gl_log.cell0) : gl_dense(fs.gl_lin);
}

bool fast_load(FastState& fs, hipStream_t st, const GlView& gl_log, const double* d_pos) {
  dbfgs_invalidate(fs);  // (a plan made in advance looked at the old data's distances)
  fs.gl_log = gl_log;
  fs.d_pos = d_pos;
  fs.e_stale = true;
  if (fs.packed) {
    if (gl_log.dense) return false;
    // linear class table = exp of the prepared log table, as gl_lin is of the dense GL
    fast_exp(st, gl_log.table, fs.cls_lin, 12);
    double t[12];
    if (hipMemcpyAsync(t, fs.cls_lin, sizeof t, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
    if (hipStreamSynchronize(st) != hipSuccess) return false;
    fs.u_lin = t[9];
    // the four classes of a called genotype: unit rows and a uniform one (est_maf's closed form)
    fs.called_table = t[0] == 1 && t[1] == 0 && t[2] == 0 && t[3] == 0 && t[4] == 1 && t[5] == 0 &&
                      t[6] == 0 && t[7] == 0 && t[8] == 1 && t[9] > 0 && t[10] == t[9] && t[11] == t[9];
    hipLaunchKernelGGL(k_fast_geno_interleave, dim3(8192), dim3(256), 0, st, gl_log.codes,
                       gl_log.cell0, fs.I, fs.S, fs.T, fs.C, fs.geno_il);
  } else {
    if (!gl_log.dense) return false;
    const double* lg = gl_log.dense + gl_log.cell0 * 3;
    fast_exp(st, lg, fs.gl_lin, fs.I * fs.S * 3);
    const uint64_t n_it = (fs.I + 31) / 32;
    hipLaunchKernelGGL(k_fast_gl_interleave, dim3((unsigned)((uint64_t)fs.C * fs.T * n_it)),
                       dim3(256), 0, st, lg, fs.I, fs.S, fs.T, fs.C,
                       reinterpret_cast<double2*>(fs.glq_il));
  }
  hipLaunchKernelGGL(k_fast_gl_scale, dim3((unsigned)((uint64_t)fs.C * ((fs.I + 31) / 32))), dim3(256),
                     0, st, gl_log, fs.I, fs.S, fs.T, fs.C, fs.gl_scale_c);
  hipLaunchKernelGGL(k_fast_pos_interleave, dim3(1024), dim3(256), 0, st, d_pos, fs.S, fs.T, fs.C,
                     fs.pos_il);
  hipLaunchKernelGGL(k_fast_chunk_scale, dim3(fs.C), dim3(64), 0, st, fs.pos_il, fs.T,
                     reinterpret_cast<double2*>(fs.chunk_scale));
  // the largest finite distance decides when the objective kernel may use its
  // small-argument exp for the alpha +- eh probes
  unsigned long long* d_m = reinterpret_cast<unsigned long long*>(fs.bound);
  if (hipMemsetAsync(d_m, 0, sizeof(unsigned long long), st) != hipSuccess) return false;
  hipLaunchKernelGGL(k_fast_max_finite, dim3(256), dim3(256), 0, st, d_pos, fs.S, d_m);
  unsigned long long bits = 0;
  if (hipMemcpyAsync(&bits, d_m, sizeof bits, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
  if (hipStreamSynchronize(st) != hipSuccess) return false;
  std::memcpy(&fs.dmax_finite, &bits, sizeof bits);
  {  // mean finite distance, from the lane-chunks' sums (padding sites are d = 0 and no sites)
    std::vector<double> cs((size_t)2 * fs.J);
    if (hipMemcpy(cs.data(), fs.chunk_scale, cs.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
      return false;
    double sum = 0, starts = 0;
    for (size_t k = 0; k < fs.J; ++k) {
      sum += cs[2 * k];
      starts += cs[2 * k + 1];
    }
    const double n_finite = (double)fs.S - starts;
    fs.alpha_small_min = (n_finite > 0 && sum > 0) ? 1e-6 / (sum / n_finite) : HUGE_VAL;
  }
  return hipGetLastError() == hipSuccess;
}

// After an allele-frequency update: the interleaved frequency table (and the MAF check);
// the emissions themselves are refreshed either by fast_refresh_emissions or, for free, by
// the next forward walk (fast_lkl_launch with fresh emissions).
bool fast_refresh_freq_table(FastState& fs, hipStream_t st, const double* d_freq, int* d_flags) {
  hipLaunchKernelGGL(k_fast_freq_interleave, dim3(1024), dim3(256), 0, st, d_freq, fs.S, fs.T, fs.C,
                     fs.freq_il, d_flags);
  fs.e_stale = true;
  return hipGetLastError() == hipSuccess;
}


bool fast_viterbi(FastState& fs, hipStream_t st, const double* d_freq, const double* d_indF,
                  const double* d_alpha, uint8_t* d_bp, uint8_t* d_path_sites, int* d_flags,
                  double* d_scratch, uint64_t chunk_sites) {
  // Decoding runs once per analysis and must give the reference's path, ties and
  // its in-place update included: use the exact-mode kernels on log emissions.  The
  // site-major log emissions [S][I][2] (16 B per site and individual; the fast path keeps
  // only the 8 B ratio) live in a buffer of their own for the duration of the call.
  double* eprob_log = nullptr;
  if (!dalloc(&eprob_log, (size_t)fs.I * fs.S * 2)) return false;
  launch_emission_exact(st, fs.gl_log, d_freq, eprob_log, fs.S, fs.I, d_flags);
  launch_viterbi_exact(st, eprob_log, fs.d_pos, fs.S, fs.I, d_indF, d_alpha, d_bp, d_path_sites,
                       d_scratch, chunk_sites, fs.sw.exact_serial != 0);
  const bool ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
  (void)hipFree(eprob_log);
  return ok;
}

bool fast_viterbi_forward(FastState& fs, hipStream_t st, const double* d_freq, const double* d_indF,
                          const double* d_alpha, uint8_t* d_bp, int* d_flags, double* d_scratch,
                          uint64_t chunk_sites, bool chain_start) {
  double* eprob_log = nullptr;
  if (!dalloc(&eprob_log, (size_t)fs.I * fs.S * 2)) return false;
  launch_emission_exact(st, fs.gl_log, d_freq, eprob_log, fs.S, fs.I, d_flags);
  launch_viterbi_fwd_exact(st, eprob_log, fs.d_pos, fs.S, fs.I, d_indF, d_alpha, d_bp, d_scratch,
                           chunk_sites, chain_start, fs.sw.exact_serial != 0);
  const bool ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
  (void)hipFree(eprob_log);
  return ok;
}

bool fast_export_emissions(FastState& fs, hipStream_t st, double* d_out) {
  hipLaunchKernelGGL(k_fast_export_e, dim3(2048), dim3(256), 0, st, lkl_arrays(fs), fs.packed,
                     fs.gl_lin, fs.I, fs.S, fs.T, fs.C, d_out);
  return hipGetLastError() == hipSuccess;
}

}  // namespace nghmm
