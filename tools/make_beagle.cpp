// make_beagle -- a synthetic BEAGLE-style genotype-likelihood file at BASELINE configs[2]'s
// encoding (header + 3 id columns + 3 x I normal-space likelihoods per line, gzip), and its
// positions file, fast enough for 10^9 cells: lines are formatted and deflated in parallel,
// in BGZF blocks (what bgzip / ANGSD write; any gzip reader sees one stream) or, with "members",
// as one plain gzip member per 256 lines.
//
//   make_beagle N_IND N_SITES OUT_PREFIX [SEED [bgzf|members]]  ->  OUT_PREFIX.beagle.gz, OUT_PREFIX.pos.gz
//
// Data model: scripts/ngsF-HMMsim.R's, simplified (site frequency 0.2, no IBD tracts: this file
// exercises the READER; the EM kernels are measured on bench.py's data): genotypes from HWE,
// depth Poisson(2), reads with error 0.01, likelihoods normalised to sum 1, six decimals.
#include <omp.h>
#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static inline uint64_t splitmix(uint64_t& s) {
  uint64_t z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static inline double unif(uint64_t& s) { return (splitmix(s) >> 11) * (1.0 / 9007199254740992.0); }

// 22 chromosomes of equal length, sites 100 bp apart, the first of each at position 1
static inline uint64_t site_pos(uint64_t s, uint64_t S) {
  const uint64_t c = s * 22 / S, first = (c * S + 21) / 22;
  return 1 + (s - first) * 100;
}

static inline char* put6(char* p, double v) {  // "%.6f" of v in [0, 1]
  long n = std::lround(v * 1e6);
  if (n > 1000000) n = 1000000;
  *p++ = (char)('0' + n / 1000000);
  *p++ = '.';
  n %= 1000000;
  for (int d = 100000; d >= 1; d /= 10) {
    *p++ = (char)('0' + n / d);
    n %= d;
  }
  return p;
}

static bool g_bgzf = true;

// BGZF (the SAM specification's block format, what bgzip and ANGSD write): gzip members of at
// most 64 KB, each with an extra field 'B','C' holding the member's size - 1
static void bgzf_block(const char* text, size_t n, std::vector<unsigned char>& out) {
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);  // raw deflate
  const size_t at = out.size(), bound = deflateBound(&zs, n);
  out.resize(at + 18 + bound + 8);
  zs.next_in = (Bytef*)text;
  zs.avail_in = (uInt)n;
  zs.next_out = out.data() + at + 18;
  zs.avail_out = (uInt)bound;
  deflate(&zs, Z_FINISH);
  const size_t c = zs.total_out, bsize = 18 + c + 8;
  deflateEnd(&zs);
  if (bsize > 65536) {
    fprintf(stderr, "BGZF block too large\n");
    exit(1);
  }
  const unsigned char head[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0,
                                  (unsigned char)((bsize - 1) & 0xff), (unsigned char)((bsize - 1) >> 8)};
  memcpy(out.data() + at, head, 18);
  const uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), (const Bytef*)text, (uInt)n), isz = (uint32_t)n;
  unsigned char* t = out.data() + at + 18 + c;
  for (int k = 0; k < 4; k++) t[k] = (crc >> (8 * k)) & 0xff, t[4 + k] = (isz >> (8 * k)) & 0xff;
  out.resize(at + bsize);
}

static std::vector<unsigned char> gz_member(const std::vector<char>& text) {
  if (g_bgzf) {
    std::vector<unsigned char> out;
    const size_t piece = 0xff00;  // bgzip's block size: the deflated block always fits 64 KB
    for (size_t o = 0; o < text.size(); o += piece)
      bgzf_block(text.data() + o, std::min(piece, text.size() - o), out);
    return out;
  }
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  deflateInit2(&zs, 1, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY);  // level 1, gzip wrapper
  std::vector<unsigned char> out(deflateBound(&zs, text.size()) + 64);
  zs.next_in = (Bytef*)text.data();
  zs.avail_in = (uInt)text.size();
  zs.next_out = out.data();
  zs.avail_out = (uInt)out.size();
  deflate(&zs, Z_FINISH);
  out.resize(zs.total_out);
  deflateEnd(&zs);
  return out;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: make_beagle N_IND N_SITES OUT_PREFIX [SEED [bgzf|members]]\n");
    return 2;
  }
  const uint64_t I = strtoull(argv[1], nullptr, 10), S = strtoull(argv[2], nullptr, 10);
  const std::string prefix = argv[3];
  const uint64_t seed = argc > 4 ? strtoull(argv[4], nullptr, 10) : 1;
  g_bgzf = !(argc > 5 && !strcmp(argv[5], "members"));  // members: plain gzip members of 256 lines
  const uint64_t per = 256;  // sites formatted and deflated per task
  const uint64_t n_blocks = (S + per - 1) / per;
  FILE* fo = fopen((prefix + ".beagle.gz").c_str(), "wb");
  if (!fo) return 1;
  {  // header: no numeric token, so the reader skips it
    std::vector<char> h;
    std::string s = "marker\tallele1\tallele2";
    for (uint64_t i = 0; i < I; i++)
      for (int k = 0; k < 3; k++) s += "\tInd" + std::to_string(i);
    s += "\n";
    h.assign(s.begin(), s.end());
    auto z = gz_member(h);
    fwrite(z.data(), 1, z.size(), fo);
  }
  const uint64_t wave = 64;  // blocks formatted and deflated at a time, written in order
  uint64_t text_bytes = 0;
  for (uint64_t b0 = 0; b0 < n_blocks; b0 += wave) {
    const uint64_t nb = (n_blocks - b0) < wave ? (n_blocks - b0) : wave;
    std::vector<std::vector<unsigned char>> z(nb);
    std::vector<uint64_t> tb(nb, 0);
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t k = 0; k < (int64_t)nb; k++) {
      const uint64_t blk = b0 + (uint64_t)k;
      std::vector<char> text;
      text.resize(per * (I * 27 + 64));
      char* p = text.data();
      for (uint64_t s = blk * per; s < S && s < (blk + 1) * per; s++) {
        uint64_t rs = seed * 0x100000001b3ull + s * 0x9e3779b97f4a7c15ull;
        p += sprintf(p, "chr%llu_%llu\tA\tC", (unsigned long long)(1 + s * 22 / S),
                     (unsigned long long)site_pos(s, S));
        for (uint64_t i = 0; i < I; i++) {
          const double f = 0.2;
          const int g = (unif(rs) < f) + (unif(rs) < f);
          int depth = 0;  // Poisson(2) by inversion
          {
            double u = unif(rs), pk = std::exp(-2.0), c = pk;
            while (u > c && depth < 20) {
              ++depth;
              pk *= 2.0 / depth;
              c += pk;
            }
          }
          int alt = 0;
          const double pa[3] = {0.01, 0.5, 0.99};
          for (int r = 0; r < depth; r++) alt += unif(rs) < pa[g];
          double l[3], sum = 0;
          for (int gg = 0; gg < 3; gg++) {
            l[gg] = std::pow(pa[gg], alt) * std::pow(1 - pa[gg], depth - alt);
            sum += l[gg];
          }
          for (int gg = 0; gg < 3; gg++) {
            *p++ = '\t';
            p = put6(p, l[gg] / sum);
          }
        }
        *p++ = '\n';
      }
      text.resize((size_t)(p - text.data()));
      tb[k] = text.size();
      z[k] = gz_member(text);
    }
    for (uint64_t k = 0; k < nb; k++) {
      fwrite(z[k].data(), 1, z[k].size(), fo);
      text_bytes += tb[k];
    }
  }
  if (g_bgzf) {  // the empty block that ends a BGZF file
    std::vector<unsigned char> eof;
    bgzf_block("", 0, eof);
    fwrite(eof.data(), 1, eof.size(), fo);
  }
  fclose(fo);
  gzFile fp = gzopen((prefix + ".pos.gz").c_str(), "wb1");
  for (uint64_t s = 0; s < S; s++)
    gzprintf(fp, "chr%llu\t%llu\n", (unsigned long long)(1 + s * 22 / S),
             (unsigned long long)site_pos(s, S));
  gzclose(fp);
  printf("%llu x %llu: %.1f MB of text\n", (unsigned long long)I, (unsigned long long)S, text_bytes / 1e6);
  return 0;
}
