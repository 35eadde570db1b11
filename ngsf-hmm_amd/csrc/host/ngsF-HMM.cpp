// ngsF-HMM (MI355X) -- C++ host of the GPU EM hot path.
//
// Keeps the reference's command line (parse_args.cpp:43-68: same long options, also
// with a single dash), its input conventions (shared/read_data.cpp) and its output
// files (.indF / .ibd / .geno, EM.cpp:293-380) byte-compatible, and runs the EM loop
// of EM.cpp:27-135 with every per-site-per-individual computation behind the C ABI of
// include/nghmm.h (HIP kernels).  Host-side work is what the reference also does once,
// outside its hot loop: argument parsing, file I/O, GL normalisation, initial values.
//
// Extra options: --mode exact|fast (default fast), --device N.  --n_threads (the
// reference's pool size) sets the host threads used for input normalisation and output
// formatting; results do not depend on it.
#include <fcntl.h>
#include <getopt.h>
#include <omp.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <charconv>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/nghmm.h"

namespace {

const char* kVersion = "1.1.0-mi355x";
constexpr double kINF = 1e15;        // shared/gen_func.hpp:15
constexpr double kEPSILON = 1e-5;    // shared/gen_func.hpp:16
constexpr size_t kBuffLen = 500000;  // shared/gen_func.hpp:17

struct Params {  // ngsF-HMM.hpp:13-52
  const char* in_geno = nullptr;
  const char* in_pos = nullptr;
  bool in_bin = false, in_lkl = false, in_loglkl = false;
  uint64_t n_ind = 0, n_sites = 0;
  bool call_geno = false;
  std::string in_freq, in_indF;
  int freq_est = 1, e_prob_calc = 1;
  // --ld_intended: --freq_est 2 / --e_prob 2 do what EM.cpp:224-263 evidently means instead of
  // what the reference does with them (abort).  Opt-in; parity unpinned (include/nghmm.h).
  bool ld_intended = false;
  bool indF_fixed = false, alpha_fixed = false;
  const char* out_prefix = nullptr;
  unsigned log = 0;
  bool log_bin = false;
  unsigned min_iters = 10, max_iters = 100;
  double min_epsilon = 1e-5;
  unsigned n_threads = 1, verbose = 1, seed = 0;
  int mode = NGHMM_MODE_FAST, device = 0;

  bool no_pack = false;          // --no_pack: never keep called genotypes as 2-bit codes
  unsigned n_starts = 1;         // --n_starts R: R replicates from seeds seed, seed+1, ... (ngsF-HMM.sh)
  bool keep_starts = false;      // --keep_starts: write every replicate's files, PREFIX.REP_rr.*
  unsigned n_gpus = 1;           // --n_gpus N: the sites split over N handles, one per device
  std::vector<int> devices;      // --devices a,b,..: the device of each handle (default 0..N-1)
  FILE* out = stdout;            // where this run's progress lines go (replicates: a buffer)
  std::string prefix;            // output prefix of this run
  std::vector<double> pos_dist;  // [S] Mb
  std::vector<double> freq, indF, alpha, ind_lkl;
  std::vector<uint8_t> path;
  double tot_lkl = 0, prev_tot_lkl = 0;
};

// The cohort on the device(s): handle r holds ALL individuals for the sites [lo[r], lo[r + 1])
// (one handle unless --n_gpus N; include/nghmm.h "a CHAIN of site shards").  The site axis is
// the one the input files are ordered by, so a block of sites read goes to the handle that owns
// it as it is.
struct Cohort {
  std::vector<nghmm_t*> hs;
  std::vector<uint64_t> lo;   // n() + 1 cuts; multiples of 16 sites except the last
  int n() const { return (int)hs.size(); }
  uint64_t sites(int r) const { return lo[r + 1] - lo[r]; }
  void cut(uint64_t S, unsigned n) {
    lo.assign(1, 0);
    for (unsigned r = 1; r < n; r++) lo.push_back(std::max<uint64_t>(S * r / n / 16 * 16, lo.back() + 1));
    lo.push_back(S);
  }
};

[[noreturn]] void fatal(const char* func, const char* msg) {  // gen_func.cpp:12-18
  fflush(stdout);
  fprintf(stderr, "\n=====\nERROR: [%s] %s\n=====\n\n", func, msg);
  fflush(stderr);
  exit(-1);
}

void warn(const char* func, const char* msg) {
  fflush(stdout);
  fprintf(stderr, "\n=======\nWARNING: [%s] %s\n=======\n\n", func, msg);
  fflush(stderr);
}

void check(int rc, const char* where) {
  if (rc == NGHMM_OK) return;
  const char* detail = nghmm_last_error();
  fatal(where, (detail && *detail) ? detail : nghmm_strerror(rc));
}

// --- text parsing: tokens on the separators, non-numeric tokens dropped (gen_func.cpp:390-417) ---
size_t split_doubles(char* line, const char* sep, std::vector<double>& out) {
  out.clear();
  char* p = line;
  while (*p) {
    p += strspn(p, sep);
    if (!*p) break;
    size_t len = strcspn(p, sep);
    char saved = p[len];
    p[len] = '\0';
    char* end = nullptr;
    double v = strtod(p, &end);
    if (end != p && *end == '\0') out.push_back(v);
    p[len] = saved;
    p += len;
  }
  return out.size();
}

void chomp(char* s) {
  size_t n = strlen(s);
  while (n && (s[n - 1] == '\n' || s[n - 1] == '\r')) s[--n] = '\0';
}

// --- the same tokens, fast: strtod is ~100 ns per number and a BEAGLE file of BASELINE
// configs[2] holds 3 x 10^9 of them.  A plain decimal token -- [sign] digits [. digits]
// [e [sign] digits] -- with at most 15 significant digits and a decimal exponent within
// +-22 is ONE correctly rounded IEEE operation on two exactly representable doubles
// (integer significand, power of ten: Clinger's fast path), i.e. bit for bit what strtod
// returns; every other token (17-digit output of printf("%.17g"), nan, inf, hex floats,
// junk) goes to strtod itself.
const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                           1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// token [p, e): true and *out if it is a number in strtod's sense covering the whole token
inline bool parse_token(char* p, char* e, double* out) {
  const char* q = p;
  bool neg = false;
  if (q < e && (*q == '-' || *q == '+')) neg = (*q++ == '-');
  uint64_t m = 0;
  int nd = 0, dec_exp = 0;
  bool any = false, fast = true;
  for (; q < e && *q >= '0' && *q <= '9'; ++q) {
    any = true;
    if (m || *q != '0') {
      if (++nd > 15) fast = false;
      else m = m * 10 + (uint64_t)(*q - '0');
    }
  }
  if (q < e && *q == '.') {
    ++q;
    for (; q < e && *q >= '0' && *q <= '9'; ++q) {
      any = true;
      if (m || *q != '0') {
        if (++nd > 15) fast = false;
        else m = m * 10 + (uint64_t)(*q - '0');
      }
      --dec_exp;
    }
  }
  if (any && fast && q < e && (*q == 'e' || *q == 'E')) {
    const char* r = q + 1;
    bool eneg = false;
    if (r < e && (*r == '-' || *r == '+')) eneg = (*r++ == '-');
    int ex = 0, ed = 0;
    for (; r < e && *r >= '0' && *r <= '9' && ed < 5; ++r, ++ed) ex = ex * 10 + (*r - '0');
    if (ed == 0 || ed >= 5) fast = false;  // "1e" / absurd exponents: strtod decides
    else {
      dec_exp += eneg ? -ex : ex;
      q = r;
    }
  }
  if (any && fast && q == e && dec_exp >= -22 && dec_exp <= 22) {
    double v = (double)m;
    if (dec_exp < 0) v /= kPow10[-dec_exp];
    else if (dec_exp > 0) v *= kPow10[dec_exp];
    *out = neg ? -v : v;
    return true;
  }
  const char saved = *e;  // the chunk is ours: terminate in place for strtod
  *e = '\0';
  char* end = nullptr;
  const double v = strtod(p, &end);
  const bool ok = (end != p && end == e);
  *e = saved;
  if (ok) *out = v;
  return ok;
}

// numeric tokens of the line [p, e) on the separators ' ' and '\t' (gen_func.cpp:390-417)
inline size_t split_doubles_fast(char* p, char* e, std::vector<double>& out) {
  out.clear();
  while (p < e) {
    while (p < e && (*p == ' ' || *p == '\t')) ++p;
    if (p >= e) break;
    char* t = p;
    while (t < e && *t != ' ' && *t != '\t') ++t;
    double v;
    if (parse_token(p, t, &v)) out.push_back(v);
    p = t;
  }
  return out.size();
}

// --- block-compressed input (BGZF: what bgzip and ANGSD write -- a series of gzip members of
// at most 64 KB of data each, every header carrying its member's size in a 'BC' extra field).
// A plain gzip stream can only be inflated by one thread, about 1 GB/s of text with zlib;
// BGZF members are independent, so a dispatcher reads the compressed file, cuts it at member
// boundaries (no inflating needed: sizes are in the headers, uncompressed sizes in the
// trailers) and hands groups of members to T inflating threads; read() delivers the text in
// file order.  Every member's CRC-32 and length are checked, as gzread would.
class BgzfSource {
 public:
  // true if the file starts with a BGZF member
  static bool detect(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    unsigned char h[18];
    const bool ok = fread(h, 1, 18, f) == 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 &&
                    (h[3] & 4) && h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0;
    fclose(f);
    return ok;
  }
  BgzfSource(const char* path, unsigned n_threads) : f_(fopen(path, "rb")) {
    // an allocation failure in a pipeline thread (a damaged file can ask for a lot) ends the
    // read with the reader's "cannot read GZip GENO file", not in std::terminate
    dispatcher_ = std::thread([this] { guarded([this] { dispatch(); }); });
    for (unsigned t = 0; t < n_threads; ++t)
      workers_.emplace_back([this] { guarded([this] { work(); }); });
  }
  ~BgzfSource() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_job_.notify_all();
    cv_out_.notify_all();
    cv_room_.notify_all();
    dispatcher_.join();
    for (auto& th : workers_) th.join();
    if (f_) fclose(f_);
  }
  // up to `want` bytes of text in file order; 0 at the end of the file, -1 on a damaged file
  long read(char* dst, size_t want) {
    size_t got = 0;
    while (got < want) {
      if (!cur_ || cur_pos_ == cur_->out.size()) {
        std::unique_lock<std::mutex> lk(mu_);
        cur_.reset();
        cv_out_.wait(lk, [&] { return stop_ || failed_ || (!order_.empty() && order_.front()->done) ||
                                      (order_.empty() && dispatched_all_); });
        if (failed_) return -1;
        if (stop_ || order_.empty()) break;
        cur_ = order_.front();
        order_.pop_front();
        cur_pos_ = 0;
        cv_room_.notify_one();
        if (cur_->bad) return -1;
      }
      const size_t n = std::min(want - got, cur_->out.size() - cur_pos_);
      if (n) memcpy(dst + got, cur_->out.data() + cur_pos_, n);
      cur_pos_ += n;
      got += n;
    }
    return (long)got;
  }

 private:
  template <class F>
  void guarded(F f) {
    try {
      f();
    } catch (const std::exception&) {
      std::lock_guard<std::mutex> lk(mu_);
      failed_ = true;
      cv_out_.notify_all();
      cv_job_.notify_all();
      cv_room_.notify_all();
    }
  }
  struct Job {
    std::vector<unsigned char> in;                    // whole members, back to back
    std::vector<uint32_t> off, csize, isize, out_off;  // per member: offset in `in`, sizes
    std::vector<char> out;
    bool done = false, bad = false;
  };
  void dispatch() {
    std::vector<unsigned char> buf;
    size_t have = 0, pos = 0;
    bool at_eof = !f_;
    auto job = std::make_shared<Job>();
    size_t job_out = 0;
    auto flush = [&]() -> bool {  // false: stopping
      if (job->off.empty()) return true;
      job->out.resize(job_out);
      std::unique_lock<std::mutex> lk(mu_);
      cv_room_.wait(lk, [&] { return stop_ || order_.size() < kMaxJobs; });
      if (stop_) return false;
      todo_.push_back(job);
      order_.push_back(job);
      cv_job_.notify_one();
      job = std::make_shared<Job>();
      job_out = 0;
      return true;
    };
    for (;;) {
      if (have - pos < 65536 + 64 && !at_eof) {  // refill: keep a whole member in the buffer
        buf.erase(buf.begin(), buf.begin() + pos);
        have -= pos;
        pos = 0;
        buf.resize(have + (4u << 20));
        const size_t got = fread(buf.data() + have, 1, buf.size() - have, f_);
        if (got < buf.size() - have) at_eof = true;
        have += got;
      }
      if (have - pos == 0) break;
      const unsigned char* h = buf.data() + pos;
      bool ok = have - pos >= 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4);
      uint32_t bsize = 0, xlen = 0;
      if (ok) {
        xlen = h[10] | (h[11] << 8);
        ok = false;
        for (uint32_t x = 0; x + 4 <= xlen && 12 + x + 4 <= have - pos;) {  // the 'BC' subfield
          const unsigned char* sf = h + 12 + x;
          const uint32_t slen = sf[2] | (sf[3] << 8);
          if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && 12 + x + 6 <= have - pos) {
            bsize = (sf[4] | (sf[5] << 8)) + 1u;
            ok = true;
            break;
          }
          x += 4 + slen;
        }
      }
      if (!ok || bsize < xlen + 20 || bsize > have - pos) {
        std::lock_guard<std::mutex> lk(mu_);
        failed_ = true;  // not BGZF all the way, or cut short
        cv_out_.notify_all();
        return;
      }
      const uint32_t isz = h[bsize - 4] | (h[bsize - 3] << 8) | (h[bsize - 2] << 16) |
                           ((uint32_t)h[bsize - 1] << 24);
      if (isz > 65536) {  // a BGZF member holds at most 64 KB: a damaged (or hostile) trailer
        std::lock_guard<std::mutex> lk(mu_);  // must not size a multi-GB buffer
        failed_ = true;
        cv_out_.notify_all();
        cv_job_.notify_all();
        return;
      }
      job->off.push_back((uint32_t)job->in.size());
      job->csize.push_back(bsize);
      job->isize.push_back(isz);
      job->out_off.push_back((uint32_t)job_out);
      job->in.insert(job->in.end(), h, h + bsize);
      job_out += isz;
      pos += bsize;
      if (job->off.size() >= kMembersPerJob && !flush()) return;
    }
    if (!flush()) return;
    std::lock_guard<std::mutex> lk(mu_);
    dispatched_all_ = true;
    cv_out_.notify_all();
    cv_job_.notify_all();
  }
  void work() {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    inflateInit2(&zs, -15);  // raw deflate: header and trailer are handled here
    for (;;) {
      std::shared_ptr<Job> job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_job_.wait(lk, [&] { return stop_ || !todo_.empty() || dispatched_all_ || failed_; });
        if (stop_ || failed_ || todo_.empty()) {
          if (stop_ || failed_ || dispatched_all_) break;
          continue;
        }
        job = todo_.front();
        todo_.pop_front();
      }
      for (size_t m = 0; m < job->off.size() && !job->bad; ++m) {
        const unsigned char* h = job->in.data() + job->off[m];
        const uint32_t xlen = h[10] | (h[11] << 8), bsize = job->csize[m];
        inflateReset(&zs);
        zs.next_in = (Bytef*)(h + 12 + xlen);
        zs.avail_in = bsize - xlen - 20;
        unsigned char spare;  // an empty member (the one that ends the file) has nowhere to write
        zs.next_out = job->isize[m] ? (Bytef*)(job->out.data() + job->out_off[m]) : &spare;
        zs.avail_out = job->isize[m] ? job->isize[m] : 1;
        const int rc = inflate(&zs, Z_FINISH);
        const uint32_t crc = h[bsize - 8] | (h[bsize - 7] << 8) | (h[bsize - 6] << 16) |
                             ((uint32_t)h[bsize - 5] << 24);
        if (rc != Z_STREAM_END || zs.total_out != job->isize[m] ||
            crc32(crc32(0, nullptr, 0), (const Bytef*)(job->out.data() + job->out_off[m]),
                  job->isize[m]) != crc)
          job->bad = true;
      }
      job->in.clear();
      job->in.shrink_to_fit();
      {
        std::lock_guard<std::mutex> lk(mu_);
        job->done = true;
      }
      cv_out_.notify_all();
    }
    inflateEnd(&zs);
  }
  static constexpr size_t kMembersPerJob = 64, kMaxJobs = 48;
  FILE* f_;
  std::mutex mu_;
  std::condition_variable cv_job_, cv_out_, cv_room_;
  std::deque<std::shared_ptr<Job>> todo_, order_;
  bool stop_ = false, failed_ = false, dispatched_all_ = false;
  std::shared_ptr<Job> cur_;
  size_t cur_pos_ = 0;
  std::thread dispatcher_;
  std::vector<std::thread> workers_;
};

// what the text reader went through, for the --verbose 2 timing line
uint64_t g_text_bytes = 0;
unsigned g_text_threads = 0, g_inflate_threads = 1;

// One piece of the text input on its way through the reader's pipeline: whole lines as
// gzgets would hand them out (at most kBuffLen - 1 characters each, shared/gen_func.hpp:17),
// then what a worker made of them.
struct TextChunk {
  std::vector<char> text;                      // len characters + one spare byte (see parse_token)
  size_t len = 0;
  std::vector<uint32_t> line_begin, line_end;  // [line]: offsets into text, newline excluded
  // per line: number of numeric fields; UINT32_MAX marks an empty line
  std::vector<uint32_t> n_fields;
  std::vector<uint8_t> bad_geno;               // a genotype > 2 on the line
  // the last I * n_geno fields of every line that has them, in line order
  std::vector<double> rows;
  std::vector<int8_t> grows;
  std::vector<uint32_t> row_of_line;           // index into rows / grows, or UINT32_MAX
  bool parsed = false;
};

// shared/read_data.cpp:165-218 + ngsF-HMM.cpp:75-86
void read_dist(Params& P) {
  gzFile fh = gzopen(P.in_pos, "r");
  if (!fh) fatal(__FUNCTION__, "cannot open POS file!");
  std::vector<char> buf(kBuffLen);
  P.pos_dist.assign(P.n_sites, INFINITY);
  std::string prev_chr;
  uint64_t prev_pos = 0, s = 0;
  while (gzgets(fh, buf.data(), (int)kBuffLen) != nullptr) {
    chomp(buf.data());
    if (buf[0] == '\0' || buf[0] == '#') continue;
    char* tab = strpbrk(buf.data(), "\t");
    if (!tab) fatal(__FUNCTION__, "wrong POS file format!");
    std::string chr(buf.data(), tab - buf.data());
    const char* pos_s = tab + 1;
    if (strtod(pos_s, nullptr) == 0) {  // header
      fprintf(stderr, "> Header found! Skipping line...\n");
      continue;
    }
    if (s >= P.n_sites) fatal(__FUNCTION__, "wrong number of lines in POS file!");
    if (prev_chr.empty()) prev_chr = chr;
    if (chr == prev_chr) {
      P.pos_dist[s] = strtod(pos_s, nullptr) - (double)prev_pos;
      if (P.pos_dist[s] < 1) fatal(__FUNCTION__, "invalid distance between adjacent sites!");
    } else {
      P.pos_dist[s] = INFINITY;
      prev_chr = chr;
    }
    prev_pos = strtoul(pos_s, nullptr, 0);
    s++;
  }
  gzclose(fh);
  if (s != P.n_sites) fatal(__FUNCTION__, "wrong number of lines in POS file!");
  for (auto& d : P.pos_dist) d /= 1e6;
}

// shared/read_data.cpp:13-116: file -> device, a block of sites at a time (the whole matrix
// never exists on the host).  The host hands over the RAW values as the reference's reader sees
// them; conversion to log space, the two normalisations and the optional genotype call
// (read_data.cpp:36-40,89-98; ngsF-HMM.cpp:101-117) run on the device
// (nghmm_load_gl_raw_sites).  A packed handle (called genotypes as 2-bit codes) takes a
// called-genotype file as plain genotype values (nghmm_load_geno_sites).  Returns NGHMM_OK, or
// NGHMM_ERR_NOT_PACKABLE when a packed handle cannot represent the input (the caller then
// repeats the load with an unpacked handle).
int load_geno(Params& P, Cohort& C, bool packed) {
  const uint64_t I = P.n_ind, S = P.n_sites;
  const uint64_t n_geno = P.in_lkl ? 3 : 1;
  double unread;
  const uint64_t unread_bits = NGHMM_GL_UNREAD_BITS;
  memcpy(&unread, &unread_bits, sizeof unread);
  const int N = C.n();
  // the distance in front of a later handle's first site is the file's (+inf only at a
  // chromosome start)
  for (int r = 0; r < N; r++) check(nghmm_load_begin(C.hs[r], P.pos_dist.data() + C.lo[r]), "read_geno");
  // a block [ns][I][..] of sites goes to the handle(s) whose range it lies in, as it is
  auto send_gl = [&](uint64_t s0, uint64_t ns, const double* blockbuf, int space, int check_nan) {
    int rc = NGHMM_OK;
    for (int r = 0; r < N && rc == NGHMM_OK; r++) {
      const uint64_t a = std::max(s0, C.lo[r]), b = std::min(s0 + ns, C.lo[r + 1]);
      if (a < b)
        rc = nghmm_load_gl_raw_sites(C.hs[r], a - C.lo[r], b - a, blockbuf + (a - s0) * I * 3, space,
                                     P.call_geno ? 1 : 0, check_nan);
    }
    return rc;
  };
  auto send_geno = [&](uint64_t s0, uint64_t ns, const int8_t* blockbuf) {
    int rc = NGHMM_OK;
    for (int r = 0; r < N && rc == NGHMM_OK; r++) {
      const uint64_t a = std::max(s0, C.lo[r]), b = std::min(s0 + ns, C.lo[r + 1]);
      if (a < b) rc = nghmm_load_geno_sites(C.hs[r], a - C.lo[r], b - a, blockbuf + (a - s0) * I);
    }
    return rc;
  };
  gzFile fh = gzopen(P.in_geno, P.in_bin ? "rb" : "r");
  if (!fh) fatal(__FUNCTION__, "cannot open GENO file!");
  gzbuffer(fh, 1 << 20);
  uint64_t block = (64ull << 20) / (I * 24);  // sites per block: about 64 MB of likelihoods
  if (const char* env = getenv("NGHMM_HOST_BLOCK_SITES")) block = strtoull(env, nullptr, 10);  // tests
  if (block < 1) block = 1;
  if (block > S) block = S;
  int rc = NGHMM_OK;
  if (P.in_bin) {
    std::vector<double> buf((size_t)block * I * 3);
    const int space = P.in_loglkl ? NGHMM_GL_LOG : NGHMM_GL_NORMAL_BINARY;
    for (uint64_t s0 = 0; s0 < S && rc == NGHMM_OK; s0 += block) {
      const uint64_t ns = (S - s0) < block ? (S - s0) : block;
      for (uint64_t s = 0; s < ns; s++) {
        const int want = (int)(I * 3 * sizeof(double));
        if (gzread(fh, &buf[s * I * 3], want) != want)
          fatal(__FUNCTION__, gzeof(fh)
                                  ? "GENO file at premature EOF. Check GENO file and number of sites!"
                                  : "cannot read binary GENO file. Check GENO file and number of sites!");
      }
      // NaN check: read_data.cpp:42-45 (binary input only)
      rc = send_gl(s0, ns, buf.data(), space, 1);
    }
  } else {
    // Text input as a pipeline: ONE thread inflates (a gzip stream is sequential) and cuts the
    // text into pieces of whole lines, W workers tokenise the pieces in parallel, and this
    // thread takes the pieces back IN FILE ORDER, applies the reader's line rules
    // (shared/read_data.cpp:52-110: headers, empty lines, field counts -- they depend on how
    // many sites have been consumed, so they stay sequential) and sends runs of sites to the
    // device.  Same values, same messages, same first error as the line-by-line loop.
    const bool as_codes = packed && !P.in_lkl;  // called genotypes straight to 2-bit codes
    const int space = (P.in_lkl && !P.in_loglkl) ? NGHMM_GL_NORMAL_TEXT : NGHMM_GL_LOG;
    size_t chunk_bytes = 8u << 20;
    if (const char* env = getenv("NGHMM_HOST_CHUNK_BYTES")) chunk_bytes = strtoull(env, nullptr, 10);  // tests
    if (chunk_bytes < 16) chunk_bytes = 16;
    unsigned W = std::thread::hardware_concurrency();
    if (W > 16) W = 16;  // an 8-GPU node runs several of these; inflate is the limit anyway
    if (W < 1) W = 1;
    g_text_threads = W;
    g_text_bytes = 0;
    const size_t max_inflight = 2 * W + 2;

    std::mutex mu;
    std::condition_variable cv_work, cv_done, cv_room;
    std::deque<std::shared_ptr<TextChunk>> todo;      // cut, not yet taken by a worker
    std::deque<std::shared_ptr<TextChunk>> in_order;  // every piece, in file order
    bool eof = false, stop = false;
    const char* read_error = nullptr;

    auto cut_lines = [&](TextChunk& ck) {  // gzgets semantics: '\n' ends a line, so does length
      const size_t n = ck.len;
      size_t b = 0;
      while (b < n) {
        const char* nl = (const char*)memchr(ck.text.data() + b, '\n', n - b);
        size_t e = nl ? (size_t)(nl - ck.text.data()) : n;
        if (e - b > kBuffLen - 1) e = b + (kBuffLen - 1);  // an overlong line comes in pieces
        size_t stop_at = e;
        while (stop_at > b && (ck.text[stop_at - 1] == '\r')) --stop_at;  // chomp
        ck.line_begin.push_back((uint32_t)b);
        ck.line_end.push_back((uint32_t)stop_at);
        b = (nl && e == (size_t)(nl - ck.text.data())) ? e + 1 : e;
      }
    };

    // a BGZF file is inflated by several threads, anything else by gzread on this one
    std::unique_ptr<BgzfSource> bgzf;
    g_inflate_threads = 1;
    if (BgzfSource::detect(P.in_geno) && !getenv("NGHMM_HOST_NO_BGZF")) {
      g_inflate_threads = W >= 8 ? 8 : W;
      if (const char* e = getenv("NGHMM_HOST_INFLATE_THREADS"))  // measurement
        g_inflate_threads = (unsigned)std::max(1, std::min(64, atoi(e)));
      bgzf.reset(new BgzfSource(P.in_geno, g_inflate_threads));
    }
    std::thread inflater([&] {
      std::vector<char> carry;  // the unfinished last line of the previous piece
      for (;;) {
        auto ck = std::make_shared<TextChunk>();
        ck->text.resize(carry.size() + chunk_bytes);
        if (!carry.empty()) memcpy(ck->text.data(), carry.data(), carry.size());
        size_t have = carry.size();
        carry.clear();
        bool at_eof = false;
        while (have < ck->text.size()) {
          const long got = bgzf ? bgzf->read(ck->text.data() + have, ck->text.size() - have)
                                : (long)gzread(fh, ck->text.data() + have,
                                               (unsigned)(ck->text.size() - have));
          if (got < 0) {
            std::lock_guard<std::mutex> lk(mu);
            read_error = "cannot read GZip GENO file. Check GENO file and number of sites!";
            eof = true;
            cv_work.notify_all();
            cv_done.notify_all();
            return;
          }
          if (got == 0) {
            at_eof = true;
            break;
          }
          have += (size_t)got;
        }
        ck->text.resize(have);
        if (!at_eof) {  // keep the unfinished last line for the next piece
          size_t cut = have;
          while (cut > 0 && ck->text[cut - 1] != '\n') --cut;
          if (cut == 0 && have >= kBuffLen - 1) cut = have - (have % (kBuffLen - 1));  // no newline in sight
          carry.assign(ck->text.begin() + cut, ck->text.end());
          ck->text.resize(cut);
        }
        ck->len = ck->text.size();
        g_text_bytes += ck->len;
        ck->text.push_back('\0');  // the spare byte parse_token may overwrite behind the last token
        if (ck->len) {
          cut_lines(*ck);
          std::unique_lock<std::mutex> lk(mu);
          cv_room.wait(lk, [&] { return stop || in_order.size() < max_inflight; });
          if (stop) return;
          todo.push_back(ck);
          in_order.push_back(ck);
          cv_work.notify_one();
        }
        if (at_eof) {
          std::lock_guard<std::mutex> lk(mu);
          eof = true;
          cv_work.notify_all();
          cv_done.notify_all();
          return;
        }
      }
    });

    auto worker = [&] {
      std::vector<double> t;
      for (;;) {
        std::shared_ptr<TextChunk> ck;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv_work.wait(lk, [&] { return stop || !todo.empty() || eof; });
          if (stop || (todo.empty() && eof)) return;
          ck = todo.front();
          todo.pop_front();
        }
        const size_t nl = ck->line_begin.size();
        ck->n_fields.assign(nl, 0);
        ck->bad_geno.assign(nl, 0);
        ck->row_of_line.assign(nl, UINT32_MAX);
        uint32_t n_rows = 0;
        for (size_t l = 0; l < nl; ++l) {
          char* b = ck->text.data() + ck->line_begin[l];
          char* e = ck->text.data() + ck->line_end[l];
          if (b == e) {
            ck->n_fields[l] = UINT32_MAX;
            continue;
          }
          const size_t nf = split_doubles_fast(b, e, t);
          ck->n_fields[l] = (uint32_t)nf;
          if (nf < I * n_geno) continue;
          const double* ptr = t.data() + (nf - I * n_geno);  // last I * n_geno columns
          ck->row_of_line[l] = n_rows;
          if (P.in_lkl) {
            ck->rows.insert(ck->rows.end(), ptr, ptr + I * 3);
          } else if (as_codes) {
            const size_t o = ck->grows.size();
            ck->grows.resize(o + I);
            for (uint64_t i = 0; i < I; i++) {
              const int gg = (int)ptr[i];
              if (gg > 2) ck->bad_geno[l] = 1;
              ck->grows[o + i] = (int8_t)(gg < 0 ? -1 : gg);
            }
          } else {
            const size_t o = ck->rows.size();
            ck->rows.resize(o + I * 3);
            for (uint64_t i = 0; i < I; i++) {
              const int gg = (int)ptr[i];
              double* g = &ck->rows[o + i * 3];
              if (gg > 2) {
                ck->bad_geno[l] = 1;
              } else if (gg >= 0) {
                g[0] = g[1] = g[2] = -kINF;  // read_data.cpp:21
                g[gg] = log(1);
              } else {
                g[0] = g[1] = g[2] = log((double)1 / 3);
              }
            }
          }
          ++n_rows;
        }
        {
          std::lock_guard<std::mutex> lk(mu);
          ck->parsed = true;
        }
        cv_done.notify_all();
      }
    };
    std::vector<std::thread> pool;
    for (unsigned w = 0; w < W; ++w) pool.emplace_back(worker);
    auto shut_down = [&] {
      {
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
      }
      cv_work.notify_all();
      cv_room.notify_all();
      inflater.join();
      for (auto& th : pool) th.join();
    };
    // errors must not leave threads behind: fatal() exits the process
    auto die = [&](const char* msg) {
      shut_down();
      fatal("load_geno", msg);
    };

    // the sites of the current run [s0, s0 + filled) on their way to the device
    std::vector<double> dbuf;
    std::vector<int8_t> gbuf;
    uint64_t s0 = 0, filled = 0, s = 0;
    auto flush = [&]() {
      if (!filled || rc != NGHMM_OK) return;
      rc = as_codes ? send_geno(s0, filled, gbuf.data()) : send_gl(s0, filled, dbuf.data(), space, 0);
      s0 += filled;
      filled = 0;
      dbuf.clear();
      gbuf.clear();
    };
    bool trailing_data = false;
    for (;;) {
      std::shared_ptr<TextChunk> ck;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return (!in_order.empty() && in_order.front()->parsed) ||
                                      (in_order.empty() && eof); });
        if (read_error) {
          lk.unlock();
          die(read_error);
        }
        if (in_order.empty()) break;  // end of file
        ck = in_order.front();
        in_order.pop_front();
        cv_room.notify_one();
      }
      if (s >= S) {  // all sites are in: anything further is "not at EOF"
        trailing_data = true;
        break;
      }
      const size_t nl = ck->line_begin.size();
      for (size_t l = 0; l < nl && rc == NGHMM_OK; ++l) {
        if (s >= S) {
          trailing_data = true;
          break;
        }
        const uint32_t nf = ck->n_fields[l];
        if (nf == UINT32_MAX) {
          // (sic) an empty line still consumes a site, as in the reference (read_data.cpp:60-61):
          // its cells stay "unread", which 2-bit codes cannot express
          if (as_codes) {
            rc = NGHMM_ERR_NOT_PACKABLE;
            break;
          }
          dbuf.insert(dbuf.end(), I * 3, unread);
        } else {
          if (!nf || (s == 0 && nf < I * n_geno)) {
            fprintf(stderr, "> Header found! Skipping line...\n");
            if (s != 0) warn("load_geno", " header found but not on first line. Is this an error?");
            continue;
          }
          if (nf < I * n_geno) die("wrong GENO file format. Less fields than expected!");
          if (ck->bad_geno[l])
            die("wrong GENO file format. Genotypes must be coded as {-1,0,1,2} !");
          const uint32_t r = ck->row_of_line[l];
          if (as_codes)
            gbuf.insert(gbuf.end(), ck->grows.begin() + (size_t)r * I,
                        ck->grows.begin() + (size_t)(r + 1) * I);
          else
            dbuf.insert(dbuf.end(), ck->rows.begin() + (size_t)r * I * 3,
                        ck->rows.begin() + (size_t)(r + 1) * I * 3);
        }
        s++;
        if (++filled == block) flush();
      }
      if (rc != NGHMM_OK || trailing_data) break;
      flush();  // a piece's sites go out together: the next piece is being tokenised meanwhile
    }
    flush();
    shut_down();
    if (rc == NGHMM_ERR_NOT_PACKABLE) {
      gzclose(fh);
      return rc;
    }
    check(rc, "read_geno");
    if (s < S)
      fatal(__FUNCTION__, "GENO file at premature EOF. Check GENO file and number of sites!");
    if (trailing_data)
      fatal(__FUNCTION__, "GENO file not at EOF. Check GENO file and number of sites!");
    gzclose(fh);
    fh = nullptr;
  }
  if (rc == NGHMM_ERR_NOT_PACKABLE) {
    if (fh) gzclose(fh);
    return rc;
  }
  check(rc, "read_geno");
  if (fh) {  // binary input (the text pipeline has done its own end-of-file check)
    char c;
    gzread(fh, &c, 1);
    if (!gzeof(fh)) fatal(__FUNCTION__, "GENO file not at EOF. Check GENO file and number of sites!");
    gzclose(fh);
  }
  for (nghmm_t* h : C.hs) {
    rc = nghmm_load_end(h);
    if (rc == NGHMM_ERR_NOT_PACKABLE) return rc;
    check(rc, "read_geno");
  }
  return rc;
}

// GSL's gsl_rng_taus (Tausworthe, L'Ecuyer 1996), the only GSL generator the reference
// uses (parse_args.cpp:232-233); restated from the published recurrence.
struct Taus {
  uint32_t s1, s2, s3;
  static uint32_t lcg(uint32_t n) { return 69069u * n; }
  uint32_t next() {
    s1 = ((s1 & 4294967294u) << 12) ^ (((s1 << 13) ^ s1) >> 19);
    s2 = ((s2 & 4294967288u) << 4) ^ (((s2 << 2) ^ s2) >> 25);
    s3 = ((s3 & 4294967280u) << 17) ^ (((s3 << 3) ^ s3) >> 11);
    return s1 ^ s2 ^ s3;
  }
  explicit Taus(uint32_t s) {
    if (s == 0) s = 1;
    s1 = lcg(s);
    s2 = lcg(s1);
    s3 = lcg(s2);
    for (int i = 0; i < 6; i++) next();
  }
  double uniform() { return next() / 4294967296.0; }
};

double clampd(double v, double lo, double hi) {
  const double a = (v >= lo) ? v : lo;  // the reference's max/min macros
  return (a <= hi) ? a : hi;
}

// parse_args.cpp:229-363 (initial indF/alpha and freq); --freq e is done on the GPU
bool init_values(Params& P, Cohort& C) {
  const uint64_t I = P.n_ind, S = P.n_sites;
  Taus rng(P.seed);
  const double f_min = 0.000001, f_max = 1 - f_min;
  P.indF.assign(I, 0);
  P.alpha.assign(I, 0);
  std::vector<double> t;
  std::vector<char> buf(kBuffLen);
  gzFile fh;
  if (P.in_indF == "r") {
    if (P.verbose >= 1) fprintf(P.out, "==> Using random initial inbreeding values.\n");
    for (uint64_t i = 0; i < I; i++) {
      P.indF[i] = f_min + rng.uniform() * (f_max - f_min);
      P.alpha[i] = f_min + rng.uniform() * (f_max - f_min);
    }
  } else if ((fh = gzopen(P.in_indF.c_str(), "r")) != nullptr) {
    if (P.verbose >= 1)
      fprintf(P.out, "==> Reading initial inbreeding values from file \"%s\".\n", P.in_indF.c_str());
    uint64_t i = 0;
    while (gzgets(fh, buf.data(), (int)kBuffLen) != nullptr) {
      chomp(buf.data());
      if (buf[0] == '\0') continue;
      if (i >= I || split_doubles(buf.data(), " ,-\t", t) != 2)
        fatal(__FUNCTION__, "wrong INDF file format!");
      P.indF[i] = clampd(t[0], f_min, f_max);
      P.alpha[i] = clampd(t[1], f_min, f_max);
      i++;
    }
    gzclose(fh);
  } else {
    if (P.verbose >= 1)
      fprintf(P.out, "==> Setting initial inbreeding values to: %s\n", P.in_indF.c_str());
    std::string tmp = P.in_indF;
    if (split_doubles(&tmp[0], ",-", t) != 2) fatal(__FUNCTION__, "wrong INDF parameters format!");
    for (uint64_t i = 0; i < I; i++) {
      P.indF[i] = clampd(t[0], f_min, f_max);
      P.alpha[i] = clampd(t[1], f_min, f_max);
    }
  }

  const double q_min = 0.01, q_max = 0.5 - q_min;
  P.freq.assign(S, q_min);
  bool estimate = false;
  if (P.in_freq == "r") {
    if (P.verbose >= 1) fprintf(P.out, "==> Using random initial frequency values.\n");
    for (uint64_t s = 0; s < S; s++) P.freq[s] = q_min + rng.uniform() * (q_max - q_min);
  } else if (P.in_freq == "e") {
    if (P.verbose >= 1) fprintf(P.out, "==> Estimating initial frequency values assuming HWE.\n");
    estimate = true;
  } else if ((fh = gzopen(P.in_freq.c_str(), "r")) != nullptr) {
    if (P.verbose >= 1)
      fprintf(P.out, "==> Reading initial frequency values from file \"%s\".\n", P.in_freq.c_str());
    uint64_t s = 0;
    while (gzgets(fh, buf.data(), (int)kBuffLen) != nullptr) {
      chomp(buf.data());
      if (buf[0] == '\0') continue;
      const size_t n = split_doubles(buf.data(), " ,-\t", t);
      if (!n) {
        fprintf(P.out, "> Header found! Skipping line...\n");
        continue;
      }
      if (s >= S || n != 1) fatal(__FUNCTION__, "wrong FREQ file format!");
      P.freq[s++] = clampd(t[0], q_min, q_max);
    }
    gzclose(fh);
  } else {
    if (P.verbose >= 1) fprintf(P.out, "==> Setting initial frequency values to: %s\n", P.in_freq.c_str());
    for (uint64_t s = 0; s < S; s++) P.freq[s] = clampd(atof(P.in_freq.c_str()), q_min, q_max);
  }
  for (int r = 0; r < C.n(); r++)
    check(nghmm_set_params(C.hs[r], P.indF.data(), P.alpha.data(), P.freq.data() + C.lo[r]),
          "init_output");
  return estimate;
}

// printf("%f") / printf("%.10f") through std::to_chars: the same correctly rounded
// digits (checked against printf on 5M values incl. ties), ~10x faster, thread-safe
inline char* put_fixed(char* p, double v, int prec) {
  if (std::isnan(v) || std::isinf(v)) return p + sprintf(p, prec == 6 ? "%f" : "%.10f", v);
  auto r = std::to_chars(p, p + 400, v, std::chars_format::fixed, prec);
  return r.ptr;
}

// The .ibd and .geno files of a large run are gigabytes (1000 x 1M: 10 GB and 24 GB) that come
// off the device a batch at a time.  Every batch has a known place in its file, so it is
// fetched into one of a few pinned buffers (nghmm_alloc_host: the copy from the device runs at
// the PCIe rate) and handed to a pool of threads that pwrite() it there, while the main thread
// fetches the next one: copies into the page cache run in parallel with each other and with the
// device's formatting.
class PositionalWriter {
 public:
  // one pool for the process (print_iter runs once per --log interval and per replicate):
  // pinning and unpinning a few hundred MB costs as much as writing a GB
  static PositionalWriter& get(size_t slot_bytes) {
    static std::mutex mu;
    static PositionalWriter* w = nullptr;   // lives until the process ends
    std::lock_guard<std::mutex> lk(mu);
    if (!w) w = new PositionalWriter(slot_bytes, 6, 4);
    if (w->cap_ < slot_bytes) fatal("print_iter", "output buffers of another size are in use!");
    return *w;
  }
  // the writes of one caller: wait() returns when all of them are done
  struct Job {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t pending = 0;
    bool failed = false;
  };
  size_t slot_bytes() const { return cap_; }
  // a free buffer (waits for one)
  int acquire() {
    std::unique_lock<std::mutex> lk(mu_);
    cv_free_.wait(lk, [&] { return !free_.empty(); });
    const int k = free_.back();
    free_.pop_back();
    return k;
  }
  char* data(int slot) { return slots_[slot]; }
  // write the first len bytes of the buffer at `offset` of fd; the buffer is free again afterwards
  void submit(Job& job, int slot, int fd, uint64_t offset, size_t len) {
    {
      std::lock_guard<std::mutex> lk(job.mu);
      ++job.pending;
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      todo_.push_back(Task{&job, slot, fd, offset, len});
    }
    cv_todo_.notify_one();
  }
  static void wait(Job& job) {
    std::unique_lock<std::mutex> lk(job.mu);
    job.cv.wait(lk, [&] { return job.pending == 0; });
    if (job.failed) fatal("print_iter", "cannot write the output files!");
  }

 private:
  PositionalWriter(size_t slot_bytes, int n_slots, int n_threads) : cap_(slot_bytes) {
    for (int k = 0; k < n_slots; k++) {
      void* p = nghmm_alloc_host(slot_bytes);
      if (!p) fatal("print_iter", "cannot allocate output buffers!");
      slots_.push_back(static_cast<char*>(p));
      free_.push_back(k);
    }
    for (int t = 0; t < n_threads; t++) std::thread([this] { work(); }).detach();
  }
  struct Task {
    Job* job;
    int slot, fd;
    uint64_t offset;
    size_t len;
  };
  void work() {
    for (;;) {
      Task t;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_todo_.wait(lk, [&] { return !todo_.empty(); });
        t = todo_.front();
        todo_.pop_front();
      }
      const char* p = slots_[t.slot];
      size_t left = t.len;
      uint64_t off = t.offset;
      bool ok = true;
      while (left) {
        const ssize_t w = pwrite(t.fd, p, left, (off_t)off);
        if (w <= 0) {
          ok = false;
          break;
        }
        p += w;
        off += (uint64_t)w;
        left -= (size_t)w;
      }
      {
        std::lock_guard<std::mutex> lk(mu_);
        free_.push_back(t.slot);
      }
      cv_free_.notify_one();
      {
        std::lock_guard<std::mutex> lk(t.job->mu);
        if (!ok) t.job->failed = true;
        --t.job->pending;
      }
      t.job->cv.notify_all();
    }
  }
  size_t cap_;
  std::vector<char*> slots_;
  std::vector<int> free_;
  std::deque<Task> todo_;
  std::mutex mu_;
  std::condition_variable cv_free_, cv_todo_;
};

// EM.cpp:293-380
void print_iter(const Params& P, Cohort& C) {
  const double t_p0 = omp_get_wtime();
  const uint64_t I = P.n_ind, S = P.n_sites;
  std::string name = P.prefix + ".indF";
  FILE* fh = fopen(name.c_str(), "w");
  if (!fh) fatal(__FUNCTION__, "cannot open INDF output file!");
  setvbuf(fh, nullptr, _IOFBF, 1 << 22);
  fprintf(fh, "%.10f\n", P.tot_lkl);
  for (uint64_t i = 0; i < I; i++) {  // the reference's uint16_t index never ends for I >= 65536
    if (P.indF[i] < kEPSILON)
      fprintf(fh, "%.5f\tNA\n", (double)0);
    else if (P.indF[i] > 1 - kEPSILON)
      fprintf(fh, "%.5f\tNA\n", (double)1);
    else
      fprintf(fh, "%.5f\t%f\n", P.indF[i], P.alpha[i]);
  }
  {
    char buf[512];
    for (uint64_t s = 0; s < S; s++) {
      char* e = put_fixed(buf, P.freq[s], 6);
      *e++ = '\n';
      fwrite(buf, 1, e - buf, fh);
    }
  }
  fclose(fh);

  // batches of about 32 MB: sites per .geno block, individuals per .ibd batch
  const uint64_t target = 32ull << 20;
  uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(S, target / (I * 24)));
  uint64_t batch = std::max<uint64_t>(1, std::min<uint64_t>(I, target / (9 * S)));
  const size_t slot = std::max<size_t>(target, std::max<size_t>(chunk * I * 24, batch * 9 * S));
  const double t_w0 = omp_get_wtime();
  PositionalWriter& W = PositionalWriter::get(slot);
  PositionalWriter::Job job;
  double t_wait = 0, t_dev = 0, t_host = 0;  // waiting for a free buffer / in device calls / host formatting
  const double t_w1 = omp_get_wtime();
  auto tick = [](double& acc, double t0) { acc += omp_get_wtime() - t0; };

  name = P.prefix + ".geno";
  const int fd_geno = open(name.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
  if (fd_geno < 0) fatal(__FUNCTION__, "cannot open GENO output file!");
  for (uint64_t s0 = 0; s0 < S; s0 += chunk) {
    const uint64_t ns = (S - s0) < chunk ? (S - s0) : chunk;
    double t = omp_get_wtime();
    const int k = W.acquire();
    tick(t_wait, t);
    t = omp_get_wtime();
    double* blk = reinterpret_cast<double*>(W.data(k));
    // EM.cpp:367-376 on the device, from the decoded path and the final frequencies; the file
    // is site-major over all individuals: a block comes from the handle(s) that own its sites
    for (int r = 0; r < C.n(); r++) {
      const uint64_t a = std::max(s0, C.lo[r]), b = std::min(s0 + ns, C.lo[r + 1]);
      if (a < b)
        check(nghmm_geno_posteriors(C.hs[r], a - C.lo[r], b - a, &blk[(a - s0) * I * 3]), "print_iter");
    }
    tick(t_dev, t);
    W.submit(job, k, fd_geno, s0 * I * 3 * sizeof(double), ns * I * 3 * sizeof(double));
  }

  name = P.prefix + ".ibd";
  const int fd_ibd = open(name.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
  if (fd_ibd < 0) fatal(__FUNCTION__, "cannot open IBD output file!");
  uint64_t at = 0;  // bytes of the file laid out so far
  {
    std::string head = "//";
    char num[64];
    for (uint64_t i = 0; i < I; i++) {
      snprintf(num, sizeof num, "\t%.10f", P.ind_lkl[i]);
      head += num;
    }
    head += '\n';
    for (size_t o = 0; o < head.size(); o += W.slot_bytes()) {
      const size_t n = std::min(W.slot_bytes(), head.size() - o);
      const int k = W.acquire();
      memcpy(W.data(k), head.data() + o, n);
      W.submit(job, k, fd_ibd, at + o, n);
    }
    at += head.size();
  }
  {  // the path lines: one digit per site
    const uint64_t per = std::max<uint64_t>(1, W.slot_bytes() / (S + 1));
    for (uint64_t i0 = 0; i0 < I; i0 += per) {
      const uint64_t nb = std::min(per, I - i0);
      double t = omp_get_wtime();
      const int k = W.acquire();
      tick(t_wait, t);
      t = omp_get_wtime();
      char* line = W.data(k);
      for (uint64_t i = 0; i < nb; i++) {
        char* l = line + i * (S + 1);
        const char* pth = reinterpret_cast<const char*>(&P.path[(i0 + i) * S]);
        for (uint64_t s = 0; s < S; s++) l[s] = (char)(pth[s] + 48);
        l[S] = '\n';
      }
      tick(t_host, t);
      W.submit(job, k, fd_ibd, at + i0 * (S + 1), nb * (S + 1));
    }
    at += I * (S + 1);
  }
  // posterior lines (EM.cpp:347-353): printf("%f") text formatted on the device, a batch of
  // individuals at a time
  {
    uint64_t widest = 0;
    for (int r = 0; r < C.n(); r++) widest = std::max(widest, C.sites(r));
    char* piece = C.n() > 1 ? static_cast<char*>(nghmm_alloc_host(batch * 9 * widest)) : nullptr;
    if (C.n() > 1 && !piece) fatal(__FUNCTION__, "cannot allocate output buffers!");
    for (uint64_t i0 = 0; i0 < I; i0 += batch) {
      const uint64_t nb = (I - i0) < batch ? (I - i0) : batch;
      double t = omp_get_wtime();
      const int k = W.acquire();
      tick(t_wait, t);
      t = omp_get_wtime();
      char* text = W.data(k);
      if (C.n() == 1) {
        check(nghmm_format_posteriors(C.hs[0], i0, nb, text), "print_iter");
      } else {
        // a line is the handles' pieces one after the other: a piece's closing newline becomes
        // the tab in front of the next piece's first value
        for (int r = 0; r < C.n(); r++) {
          const uint64_t Sr = C.sites(r);
          check(nghmm_format_posteriors(C.hs[r], i0, nb, piece), "print_iter");
          for (uint64_t i = 0; i < nb; i++) {
            char* dst = &text[(i * S + C.lo[r]) * 9];
            memcpy(dst, &piece[i * 9 * Sr], 9 * Sr);
            if (r + 1 < C.n()) dst[9 * Sr - 1] = '\t';
          }
        }
      }
      tick(t_dev, t);
      W.submit(job, k, fd_ibd, at + i0 * 9 * S, nb * 9 * S);
    }
    if (piece) nghmm_free_host(piece);
  }
  const double t_w2 = omp_get_wtime();
  PositionalWriter::wait(job);
  if (P.verbose >= 2)  // (not a line of the reference's)
    fprintf(P.out, "> output: buffers %.2f s, device calls %.2f s, path lines %.2f s, waiting for a free "
            "buffer %.2f s, last writes %.2f s\n", t_w1 - t_w0, t_dev, t_host, t_wait, omp_get_wtime() - t_w2);
  const double t_c0 = omp_get_wtime();
  if (close(fd_geno) != 0 || close(fd_ibd) != 0) fatal(__FUNCTION__, "cannot write the output files!");
  if (P.verbose >= 2) fprintf(P.out, "> output: close %.2f s, before the buffers %.2f s\n", omp_get_wtime() - t_c0, t_w0 - t_p0);
}

void sync_outputs(Params& P, Cohort& C, bool with_viterbi) {
  P.path.resize((size_t)P.n_ind * P.n_sites, 0);
  // indF / alpha are the cohort's on every handle; the frequencies those of its own sites
  for (int r = 0; r < C.n(); r++)
    check(nghmm_get_params(C.hs[r], r == 0 ? P.indF.data() : nullptr, r == 0 ? P.alpha.data() : nullptr,
                           P.freq.data() + C.lo[r]),
          "print_iter");
  if (with_viterbi) check(nghmm_chain_viterbi(C.hs.data(), C.n(), P.path.data()), "viterbi");
}

void parse_cmd_args(Params& P, int argc, char** argv) {  // parse_args.cpp:41-225
  static struct option long_options[] = {
      {"geno", required_argument, nullptr, 'g'},      {"pos", required_argument, nullptr, 'Z'},
      {"lkl", no_argument, nullptr, 'l'},             {"loglkl", no_argument, nullptr, 'L'},
      {"n_ind", required_argument, nullptr, 'n'},     {"n_sites", required_argument, nullptr, 's'},
      {"call_geno", no_argument, nullptr, 'G'},       {"freq", required_argument, nullptr, 'f'},
      {"freq_est", required_argument, nullptr, 'F'},  {"e_prob", required_argument, nullptr, 'e'},
      {"indF", required_argument, nullptr, 'i'},      {"indF_fixed", no_argument, nullptr, 'I'},
      {"alpha_fixed", no_argument, nullptr, 'A'},     {"out", required_argument, nullptr, 'o'},
      {"log", required_argument, nullptr, 'X'},       {"log_bin", required_argument, nullptr, 'b'},
      {"min_iters", required_argument, nullptr, 'm'}, {"max_iters", required_argument, nullptr, 'M'},
      {"min_epsilon", required_argument, nullptr, 'E'},
      {"n_threads", required_argument, nullptr, 'x'}, {"verbose", required_argument, nullptr, 'V'},
      {"seed", required_argument, nullptr, 'S'},      {"mode", required_argument, nullptr, 1000},
      {"device", required_argument, nullptr, 1001},   {"taus_kat", required_argument, nullptr, 1002},
      {"no_pack", no_argument, nullptr, 1003},        {"n_starts", required_argument, nullptr, 1004},
      {"keep_starts", no_argument, nullptr, 1005},    {"n_gpus", required_argument, nullptr, 1006},
      {"devices", required_argument, nullptr, 1007},  {"ld_intended", no_argument, nullptr, 1008},
      {"parse_kat", no_argument, nullptr, 1009},
      {0, 0, 0, 0}};
  long taus_kat = 0;
  bool parse_kat = false;
  P.seed = rand() % 1000;  // parse_args.cpp:30 (unseeded rand(): a constant)
  int c;
  while ((c = getopt_long_only(argc, argv, "g:Z:lLn:s:Gf:F:e:i:IAo:X:b:m:M:E:x:V:S:", long_options,
                               nullptr)) != -1)
    switch (c) {
      case 'g': P.in_geno = optarg; break;
      case 'Z': P.in_pos = optarg; break;
      case 'l': P.in_lkl = true; break;
      case 'L': P.in_lkl = true; P.in_loglkl = true; break;
      case 'n': P.n_ind = atoi(optarg); break;
      case 's': P.n_sites = atoi(optarg); break;
      case 'G': P.call_geno = true; break;
      case 'f': P.in_freq = optarg; break;
      case 'F': P.freq_est = atoi(optarg); break;
      case 'e': P.e_prob_calc = atoi(optarg); break;
      case 'i': P.in_indF = optarg; break;
      case 'I': P.indF_fixed = true; break;
      case 'A': P.alpha_fixed = true; break;
      case 'o': P.out_prefix = optarg; break;
      case 'X': P.log = atoi(optarg); break;
      case 'b': P.log = atoi(optarg); P.log_bin = true; break;
      case 'm': P.min_iters = atoi(optarg); break;
      case 'M': P.max_iters = atoi(optarg); break;
      case 'E': P.min_epsilon = atof(optarg); break;
      case 'x': P.n_threads = atoi(optarg); break;
      case 'V': P.verbose = atoi(optarg); break;
      case 'S': P.seed = atoi(optarg); break;
      case 1008: P.ld_intended = true; break;
      case 1009: parse_kat = true; break;
      case 1000:
        if (!strcmp(optarg, "exact")) P.mode = NGHMM_MODE_EXACT;
        else if (!strcmp(optarg, "fast")) P.mode = NGHMM_MODE_FAST;
        else fatal(__FUNCTION__, "invalid --mode (exact|fast)!");
        break;
      case 1001: P.device = atoi(optarg); break;
      case 1002: taus_kat = atol(optarg); break;
      case 1003: P.no_pack = true; break;
      case 1004: P.n_starts = (unsigned)atoi(optarg); break;
      case 1005: P.keep_starts = true; break;
      case 1006: P.n_gpus = (unsigned)atoi(optarg); break;
      case 1007:
        for (const char* q = optarg; *q;) {
          P.devices.push_back(atoi(q));
          q += strcspn(q, ",");
          if (*q == ',') q++;
        }
        break;
      default: exit(-1);
    }
  if (parse_kat) {  // the fast tokenizer against strtod's, line by line on standard input
    std::vector<char> line(kBuffLen);
    std::vector<double> a, b;
    unsigned long n_tok = 0, n_line = 0;
    while (fgets(line.data(), (int)kBuffLen, stdin)) {
      chomp(line.data());
      const size_t len = strlen(line.data());
      std::vector<char> copy(line.data(), line.data() + len + 1);
      split_doubles(line.data(), " \t", a);
      split_doubles_fast(copy.data(), copy.data() + len, b);
      if (a.size() != b.size() || (a.size() && memcmp(a.data(), b.data(), a.size() * sizeof(double)))) {
        printf("parse_kat MISMATCH on line %lu: %s\n", n_line + 1, line.data());
        exit(1);
      }
      n_tok += a.size();
      ++n_line;
    }
    printf("parse_kat ok %lu tokens %lu lines\n", n_tok, n_line);
    exit(0);
  }
  if (taus_kat > 0) {  // known-answer check of the generator: the N-th raw output for --seed
    Taus rng(P.seed);
    uint32_t v = 0;
    for (long k = 0; k < taus_kat; k++) v = rng.next();
    printf("%u\n", v);
    exit(0);
  }
  if (P.in_freq.empty()) P.in_freq = "r";
  if (P.in_indF.empty()) P.in_indF = "0.01-0.001";
  if (P.verbose >= 1) {
    printf("==> Input Arguments:\n");
    printf("\tgeno: %s\n\tpos: %s\n\tlkl: %s\n\tloglkl: %s\n\tn_ind: %lu\n\tn_sites: %lu\n\tcall_geno: "
           "%s\n\tfreq: %s\n\tfreq_est: %d\n\te_prob: %d\n\tindF: %s\n\tindF_fixed: %s\n\talpha_fixed: "
           "%s\n\tout: %s\n\tlog: %u\n\tlog_bin: %s\n\tmin_iters: %d\n\tmax_iters: %d\n\tmin_epsilon: "
           "%.10f\n\tn_threads: %d\n\tverbose: %d\n\tseed: %d\n\tversion: %s (%s @ %s)\n\n",
           P.in_geno, P.in_pos, P.in_lkl ? "true" : "false", P.in_loglkl ? "true" : "false",
           (unsigned long)P.n_ind, (unsigned long)P.n_sites, P.call_geno ? "true" : "false",
           P.in_freq.c_str(), P.freq_est, P.e_prob_calc, P.in_indF.c_str(),
           P.indF_fixed ? "true" : "false", P.alpha_fixed ? "true" : "false", P.out_prefix, P.log,
           P.log_bin ? "true" : "false", P.min_iters, P.max_iters, P.min_epsilon, P.n_threads,
           P.verbose, P.seed, kVersion, __DATE__, __TIME__);
  }
  if (!P.in_geno) fatal(__FUNCTION__, "genotype input file (--geno) missing!");
  if (!P.in_pos) fatal(__FUNCTION__, "positions input file (--pos) missing!");
  if (P.n_ind == 0) fatal(__FUNCTION__, "number of individuals (--n_ind) missing!");
  if (P.n_sites == 0) fatal(__FUNCTION__, "number of sites (--n_sites) missing!");
  if (P.call_geno && !P.in_lkl) fatal(__FUNCTION__, "can only call genotypes from likelihoods!");
  if (P.freq_est < 0 || P.freq_est > 2) fatal(__FUNCTION__, "invalid MAF estimation method!");
  if (P.e_prob_calc < 0 || P.e_prob_calc > 2)
    fatal(__FUNCTION__, "invalid emission probability calculation method!");
  if (P.e_prob_calc > 1)
    warn(__FUNCTION__,
         "calculation of emission probabilities accounting for LD is still under development!");
  if (!P.out_prefix) fatal(__FUNCTION__, "output prefix (--out) missing!");
  if (P.min_iters < 1 || P.max_iters < 1 || P.min_iters >= P.max_iters)
    fatal(__FUNCTION__, "invalid number of iterations!");
  if (P.n_threads < 1) fatal(__FUNCTION__, "invalid number of threads!");
  if (P.n_starts < 1) fatal(__FUNCTION__, "invalid number of starts (--n_starts)!");
  if (!P.devices.empty() && P.n_gpus == 1) P.n_gpus = (unsigned)P.devices.size();
  if (P.n_gpus < 1 || (!P.devices.empty() && P.devices.size() != P.n_gpus))
    fatal(__FUNCTION__, "invalid --n_gpus / --devices!");
  if (P.devices.empty())
    for (unsigned r = 0; r < P.n_gpus; r++) P.devices.push_back(P.n_gpus == 1 ? P.device : (int)r);
  if (P.n_gpus > 1) {
    if (P.n_sites < 32ull * P.n_gpus)
      fatal(__FUNCTION__, "too few sites for --n_gpus (the site axis is what the GPUs share)!");
    if (P.mode != NGHMM_MODE_FAST) fatal(__FUNCTION__, "--n_gpus > 1 needs --mode fast!");
    if (P.n_starts > 1) fatal(__FUNCTION__, "--n_starts and --n_gpus > 1 cannot be combined!");
  }
  P.prefix = P.out_prefix;
}

// One EM analysis on a loaded handle: initial values, the loop of EM.cpp:27-103.
void run_em(Params& P, Cohort& C) {
  const double t_init = omp_get_wtime();
  const bool estimate_freq = init_values(P, C);
  if (estimate_freq)  // --freq e: est_maf with F = 0 (parse_args.cpp:312-318); posteriors are still 0
    check(nghmm_chain_mstep_freq(C.hs.data(), C.n(), 1), "init_output");
  if (P.verbose >= 1) fprintf(P.out, "==> Calculating initial emission probabilities\n");
  for (nghmm_t* h : C.hs) check(nghmm_emission(h), "calc_emission");
  if (P.verbose >= 2) fprintf(P.out, "> initial values and emissions in %.2f s\n", omp_get_wtime() - t_init);

  // ---- EM.cpp:27-103 ----
  const uint64_t I = P.n_ind;
  P.ind_lkl.assign(I, -INFINITY);
  std::vector<double> prev_ind_lkl(I, -INFINITY), eps(I, -INFINITY);
  double max_lkl_epsilon = -INFINITY;
  uint64_t iter = 0;
  const double t_loop = omp_get_wtime();
  while ((P.prev_tot_lkl - P.tot_lkl > P.min_epsilon || max_lkl_epsilon > P.min_epsilon ||
          iter < P.min_iters) &&
         iter < P.max_iters) {
    if (P.log && (iter == 1 || iter % P.log == 0)) {
      if (P.verbose >= 1) fprintf(P.out, "==> Printing current iteration parameters\n");
      sync_outputs(P, C, false);
      print_iter(P, C);
    }
    const time_t iter_start = time(nullptr);
    iter++;
    if (P.verbose >= 1) fprintf(P.out, "\nIteration %lu:\n", (unsigned long)iter);
    if (P.verbose >= 1)
      fprintf(P.out, "==> Forward Recursion\n==> Backward Recursion\n==> Marginal probabilities\n");
    // one call per iteration: in fast mode the E-step and the M-step share their first pass
    // over the data (nghmm_iter_em); the phase lines of EM.cpp:147-272 are printed up front
    if (P.verbose >= 1) {
      if (P.indF_fixed && P.alpha_fixed)
        fprintf(P.out, "==> Inbreeding and transition parameter not estimated!\n");
      else
        fprintf(P.out, "==> Update inbreeding and transition parameter\n");
      if (P.freq_est == 0)
        fprintf(P.out, "==> Alelle frequencies not estimated!\n");
      else
        fprintf(P.out, "==> Estimating allele frequencies and calculating emission probabilities\n");
    }
    nghmm_mstep_stats stats;
    int freq_step = P.freq_est;
    if (P.ld_intended && (P.freq_est == 2 || P.e_prob_calc == 2) && P.freq_est != 0)
      freq_step |= NGHMM_LD_INTENDED | (P.e_prob_calc == 2 ? NGHMM_EPROB_LD : 0);
    check(nghmm_chain_iter_em(C.hs.data(), C.n(), freq_step, P.indF_fixed, P.alpha_fixed,
                              P.ind_lkl.data(), &stats),
          "iter_EM");
    P.prev_tot_lkl = P.tot_lkl;
    P.tot_lkl = 0;
    for (uint64_t i = 0; i < I; i++) {
      P.tot_lkl += P.ind_lkl[i];
      eps[i] = (P.ind_lkl[i] - prev_ind_lkl[i]) / fabs(prev_ind_lkl[i]);
    }
    uint64_t best = 0;  // array_max_pos, gen_func.cpp:73-84
    double mx = -INFINITY;
    for (uint64_t i = 0; i < I; i++)
      if (eps[i] > mx) { best = i; mx = eps[i]; }
    max_lkl_epsilon = eps[best];
    prev_ind_lkl = P.ind_lkl;
    const time_t iter_end = time(nullptr);
    if (P.verbose >= 1)
      fprintf(P.out, "\tLogLkl: %.15f\t max lkl epsilon: %.15f\ttime: %.0f (s)\n", P.tot_lkl,
             max_lkl_epsilon, difftime(iter_end, iter_start));
    if (P.verbose >= 3)
      for (uint64_t i = 0; i < I; i++)
        fprintf(P.out, "\tInd %lu: %.15f\t lkl epsilon: %.15f%s\n", (unsigned long)(i + 1), P.ind_lkl[i],
               eps[i], i == best ? " (max)" : "");
    fflush(P.out);
  }
  if (iter >= P.max_iters)
    fprintf(P.out, "WARN: Maximum number of iterations reached! Check if analysis converged... \n");
  if (P.verbose >= 2)  // (not a line of the reference's)
    fprintf(P.out, "> %lu EM iterations in %.3f s\n", (unsigned long)iter, omp_get_wtime() - t_loop);
}

// EM.cpp:105-127: decoding and the three output files
void finish_run(Params& P, Cohort& C) {
  if (P.verbose >= 1) fprintf(P.out, "\n==> Decoding most probable path (Viterbi)\n");
  const double t0 = omp_get_wtime();
  sync_outputs(P, C, true);
  const double t1 = omp_get_wtime();
  if (P.verbose >= 1) {
    fprintf(P.out, "Final logLkl: %f\n", P.tot_lkl);
    fprintf(P.out, "Printing final results\n");
  }
  print_iter(P, C);
  if (P.verbose >= 2)  // (not a line of the reference's)
    fprintf(P.out, "> decoded in %.2f s, output files written in %.2f s\n", t1 - t0, omp_get_wtime() - t1);
}

}  // namespace

int main(int argc, char** argv) {
  Params P;
  parse_cmd_args(P, argc, argv);
  if (P.n_threads > P.n_ind) {  // ngsF-HMM.cpp:36-39
    warn(__FUNCTION__, "adjusting threads (--n_threads) to match number of individuals!");
    P.n_threads = (unsigned)P.n_ind;
  }
  omp_set_num_threads((int)P.n_threads);
  struct stat st;  // ngsF-HMM.cpp:47-63
  if (stat(P.in_geno, &st) != 0) fatal(__FUNCTION__, "cannot check GENO file size!");
  const char* dot = strrchr(P.in_geno, '.');
  if (dot && strcmp(dot, ".gz") == 0) {
    if (P.verbose >= 1) printf("==> GZIP input file (not BINARY)\n");
    P.in_bin = false;
  } else {
    if (P.verbose >= 1) printf("==> BINARY input file (always lkl)\n");
    P.in_bin = true;
    P.in_lkl = true;
    if (P.n_sites != (uint64_t)st.st_size / sizeof(double) / P.n_ind / 3)
      fatal(__FUNCTION__, "invalid/corrupt genotype input file!");
  }
  if (P.verbose >= 1) {
    printf("==> Reading data\n");
    printf("> Sites coordinates\n");
  }
  read_dist(P);
  // The reference dies inside iter_EM for these (EM.cpp:235-238 -> gen_func.cpp:1030-1031);
  // same message, same exit code, before any GPU work.
  if ((P.freq_est == 2 || P.e_prob_calc == 2) && !P.ld_intended)
    fatal("haplo_freq", "invalid allele frequencies");
  if (P.ld_intended && (P.freq_est == 2 || P.e_prob_calc == 2))
    warn(__FUNCTION__,
         "--ld_intended: --freq_est 2 / --e_prob 2 as the code intends them; the reference "
         "aborts here, so these results have no reference counterpart (parity unpinned)");

  if (P.verbose >= 1) printf("> GENO data\n");
  // Called genotypes (a called-genotype file, or --call_geno) are four values per cell and are
  // kept as 2-bit codes; results are those of the 24-byte likelihoods (bit for bit in exact
  // mode).  An input the codes cannot express (an empty text line) falls back to likelihoods.
  bool packed = (P.call_geno || !P.in_lkl) && !P.no_pack;
  Cohort C;
  C.cut(P.n_sites, P.n_gpus);
  auto create_all = [&](bool pk) {
    for (nghmm_t* h : C.hs) nghmm_destroy(h);
    C.hs.assign(P.n_gpus, nullptr);
    for (unsigned r = 0; r < P.n_gpus; r++)
      check(nghmm_create(&C.hs[r], P.n_ind, C.sites((int)r), P.devices[r],
                         P.mode | (pk ? NGHMM_GENO_PACKED : 0)),
            "nghmm_create");
  };
  create_all(packed);
  const double t_read0 = omp_get_wtime();
  if (load_geno(P, C, packed) == NGHMM_ERR_NOT_PACKABLE) {
    packed = false;
    create_all(false);
    check(load_geno(P, C, false), "read_geno");
  }
  if (P.verbose >= 2) {  // (not a line of the reference's)
    const double dt = omp_get_wtime() - t_read0;
    if (g_text_bytes)
      printf("> GENO data on the device in %.2f s: %.1f MB of text inflated (%u thread%s), tokenised "
             "(%u threads) and loaded at %.0f MB/s\n", dt, g_text_bytes / 1e6, g_inflate_threads,
             g_inflate_threads > 1 ? "s: BGZF" : "", g_text_threads, g_text_bytes / 1e6 / dt);
    else
      printf("> GENO data on the device in %.2f s\n", dt);
  }
  // several handles: the chain's exchange (one handle: nothing to set up)
  check(nghmm_chain_setup(C.hs.data(), C.n()), "nghmm_chain_setup");
  nghmm_t* h = C.hs[0];
  if (P.n_starts == 1) {
    run_em(P, C);
    finish_run(P, C);
  } else {
    // Multi-start (ngsF-HMM.sh:77-101: N_REP replicates, each from its own seed, the one with
    // the largest final log-likelihood is kept).  The replicates share the likelihoods on the
    // device (nghmm_create_replica) and run concurrently, one host thread and one HIP stream
    // each; replicate r uses seed + r (the script draws its seeds from bash's $RANDOM).  Only
    // the best replicate is decoded and written, unless --keep_starts.
    const unsigned R = P.n_starts;
    std::vector<Params> runs(R, P);
    std::vector<nghmm_t*> hs(R, nullptr);
    std::vector<char*> bufs(R, nullptr);
    std::vector<size_t> lens(R, 0);
    hs[0] = h;
    for (unsigned r = 0; r < R; r++) {
      if (r) check(nghmm_create_replica(&hs[r], h), "nghmm_create_replica");
      runs[r].seed = P.seed + r;
      char tag[32];
      snprintf(tag, sizeof tag, ".REP_%02u", r + 1);
      runs[r].prefix = std::string(P.out_prefix) + tag;
      runs[r].out = open_memstream(&bufs[r], &lens[r]);
    }
    std::vector<Cohort> cs(R);
    for (unsigned r = 0; r < R; r++) {
      cs[r].hs = {hs[r]};
      cs[r].cut(P.n_sites, 1);
    }
    std::vector<std::thread> th;
    for (unsigned r = 0; r < R; r++) th.emplace_back([&, r] { run_em(runs[r], cs[r]); });
    for (auto& t : th) t.join();
    unsigned best = 0;
    for (unsigned r = 1; r < R; r++)
      if (runs[r].tot_lkl > runs[best].tot_lkl) best = r;   // ties: the first, like sort's order
    for (unsigned r = 0; r < R; r++) {
      if (r == best || P.keep_starts) {
        if (r == best && !P.keep_starts) runs[r].prefix = P.out_prefix;
        finish_run(runs[r], cs[r]);
      }
      fclose(runs[r].out);
      if (P.verbose >= 1) printf("\n========== Replicate %u (seed %u) ==========\n", r + 1, runs[r].seed);
      fwrite(bufs[r], 1, lens[r], stdout);
      free(bufs[r]);
    }
    if (P.keep_starts) {  // the script moves the best replicate's files to the output prefix
      for (const char* ext : {".indF", ".ibd", ".geno"}) {
        const std::string from = runs[best].prefix + ext, to = std::string(P.out_prefix) + ext;
        FILE* a = fopen(from.c_str(), "rb");
        FILE* b = fopen(to.c_str(), "wb");
        if (!a || !b) fatal(__FUNCTION__, "cannot copy the best replicate's files!");
        std::vector<char> blk(1 << 22);
        size_t n;
        while ((n = fread(blk.data(), 1, blk.size(), a)) > 0) fwrite(blk.data(), 1, n, b);
        fclose(a);
        fclose(b);
      }
    }
    printf("Best replicate: %u (seed %u), logLkl %.10f\n", best + 1, runs[best].seed,
           runs[best].tot_lkl);
    for (unsigned r = R; r-- > 1;) nghmm_destroy(hs[r]);
  }
  if (P.verbose >= 1) printf("Freeing memory...\n");
  for (nghmm_t* hh : C.hs) nghmm_destroy(hh);
  if (P.verbose >= 1) printf("Done!\n");
  return 0;
}
