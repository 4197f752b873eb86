// Runtime-compiled sum-check round kernels for general expressions.
//
// The register-program interpreter (kernels_expr.hip sc_round_prog_kernel) pays for every instruction a decode, an
// LDS round trip of its operands and a fresh extrapolation of every table operand; that leaves the multiplier at
// ~0.4 of its peak.  For large sum-checks the same program is turned into straight-line HIP source here - registers
// are C++ locals, a table's value at the wave's evaluation point is computed once and reused, the instruction stream
// is gone - and compiled for gfx950 with hiprtc (the reference's counterpart: the per-expression `indexed_calculations`
// of util/expression/evaluator.rs:135-323, which the Rust compiler specialises at build time because the expression
// is a generic parameter there; here it is data, so the specialisation happens when the expression arrives).
// One module per distinct program (hash of the code words, register count, degree), cached for the process - and its code
// object on disk for the next process (disk_* below).
#include <hip/hiprtc.h>
#include <fcntl.h>
#include <hip/hip_version.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>
#include "dev.hpp"
#include "jit_sources.inc"  // JIT_FF_CUH, JIT_FF_COLS_INC, JIT_REDUCE_CUH: the text of ff.cuh / ff_cols.inc / reduce.cuh (Makefile)

namespace lh {

struct JitKernel {
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  unsigned threads = 0, degree = 0;
  unsigned blocks_per_cu = 1;  // resident workgroups per CU (occupancy API): the grid is sized to one resident wave of them
  bool failed = false;
};

namespace {
struct JitArgs {  // mirrored in the generated source
  const Fr* in[SC_MAX_TABLES];
  const Fr* consts;
  unsigned long long size;
  Fr* partials;
  ScFinishArgs fin;
};

std::string operand(uint32_t kind, uint32_t idx) {
  std::ostringstream o;
  if (kind == PROG_REG) o << "r" << idx;
  else if (kind == PROG_CONST) o << "a.consts[" << idx << "]";
  else if (kind == PROG_PAIR) o << "p" << idx;
  else o << "t" << idx;
  return o.str();
}

std::string generate(const uint32_t* code, size_t num_instrs, uint32_t num_regs, uint32_t result_reg, int D) {
  std::ostringstream s;
  s << "#include \"ff.cuh\"\n#include \"reduce.cuh\"\nusing namespace lh;\n"
       "struct Fin { unsigned* ticket; unsigned last_ticket; Fr* out_host; unsigned* flag; unsigned seq; unsigned long long* wide; unsigned tag; unsigned long long* lanes; };\n"
       "struct Args { const Fr* in["
    << SC_MAX_TABLES
    << "]; const Fr* consts; unsigned long long size; Fr* partials; Fin fin; };\n"
       "// value of a bound table at the wave's evaluation point X = xm1 + 1: hi + (X - 1)(hi - lo)\n"
       "static __device__ __forceinline__ Fr at_x(const Fr* __restrict__ t, unsigned long long b, int xm1) {\n"
       "  const Fr lo = t[2 * b], hi = t[2 * b + 1];\n"
       "  Fr v = hi;\n"
       "  if (xm1 > 0) {\n"
       "    const Fr step = sub(hi, lo);\n"
       "    for (int k = 0; k < xm1; k++) v = add(v, step);\n"
       "  }\n"
       "  return v;\n"
       "}\n"
       // Workgroups of 4 waves (one per SIMD) working on 4 different groups of 64 pairs at the same evaluation point
       // X = blockIdx.x + 1: the straight-line code is several times the instruction cache, and waves that run it in
       // step fetch it once.  (A workgroup of D waves, one per X, would be limited to one per CU by the registers.)
       // X is the fast grid dimension so that the D workgroups reading the same pairs run at the same time and share
       // the tables through L2.
       "extern \"C\" __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void sc_round_jit(Args a) {\n"
       "  __shared__ int is_last;\n"
       "  const int D = gridDim.x, NW = 4;\n"
       "  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wave = blockIdx.x;  // wave = X - 1\n"
       "  const unsigned long long group = (unsigned long long)blockIdx.y * NW + w, ngroups = (unsigned long long)gridDim.y * NW;\n"
       "  Fr acc = Fr::zero();\n"
       "  for (unsigned long long base = group * 64; base < a.size; base += ngroups * 64) {\n"
       "    const unsigned long long b = base + lane;\n"
       "    if (b < a.size) {\n";
  for (uint32_t r = 0; r < num_regs; r++) s << "      Fr r" << r << ";\n";
  std::map<uint32_t, bool> loaded, loaded_pair;
  auto need = [&](uint32_t kind, uint32_t idx) {
    if (kind == PROG_PAIR && !loaded_pair[idx]) {  // one entry per pair, the same at every evaluation point
      loaded_pair[idx] = true;
      s << "      const Fr p" << idx << " = a.in[" << idx << "][b];\n";
    }
    if (kind != PROG_ATOM || loaded[idx]) return;
    loaded[idx] = true;
    s << "      const Fr t" << idx << " = at_x(a.in[" << idx << "], b, wave);\n";
  };
  // Sums of products share Montgomery reductions (ff.cuh dot: K x 64 + 65 multiply-adds instead of K x 129): a product is
  // not emitted where the program has it but kept PENDING in its destination register - as long as additions and
  // subtractions only combine pending registers the terms pile up (up to JIT_DOT_MAX per reduction), and the register is
  // materialised (one product, or one dot) when something else reads it, when one of the terms' operand registers is about
  // to be overwritten, or at the end.
  const bool fuse = true;
  constexpr size_t JIT_DOT_MAX = 4;
  struct Term {
    bool neg;
    uint32_t ak, ai, bk, bi;
  };
  std::vector<std::vector<Term>> pending(16);
  auto reads = [](const Term& t, uint32_t reg) { return (t.ak == PROG_REG && t.ai == reg) || (t.bk == PROG_REG && t.bi == reg); };
  // Materialising a register WRITES its variable: every other register whose terms still name that variable's old value
  // goes first.  The value itself is computed into a temporary before that (two registers may name each other's old
  // values: each needs the other's variable untouched while it is computed).
  std::function<void(uint32_t)> flush;
  size_t num_tmp = 0;
  auto before_write = [&](uint32_t reg) {
    for (uint32_t q = 0; q < 16; q++) {
      if (q == reg) continue;
      for (const Term& t : pending[q])
        if (reads(t, reg)) {
          flush(q);
          break;
        }
    }
  };
  flush = [&](uint32_t reg) {
    if (pending[reg].empty()) return;
    const std::vector<Term> ts = pending[reg];
    pending[reg].clear();
    const size_t K = ts.size(), id = num_tmp++;
    if (K == 1) {
      const Term& t = ts[0];
      s << "      const Fr f" << id << " = " << (t.neg ? "sub(Fr::zero(), " : "") << "mul(" << operand(t.ak, t.ai) << ", "
        << operand(t.bk, t.bi) << ")" << (t.neg ? ")" : "") << ";\n";
    } else {
      s << "      Fr f" << id << ";\n      {\n        const Fr xa[" << K << "] = {";
      for (size_t k = 0; k < K; k++)
        s << (k ? ", " : "") << (ts[k].neg ? "sub(Fr::zero(), " : "") << operand(ts[k].ak, ts[k].ai) << (ts[k].neg ? ")" : "");
      s << "};\n        const Fr xb[" << K << "] = {";
      for (size_t k = 0; k < K; k++) s << (k ? ", " : "") << operand(ts[k].bk, ts[k].bi);
      s << "};\n        f" << id << " = dot<FrParams, " << K << ">(xa, xb);\n      }\n";
    }
    before_write(reg);
    s << "      r" << reg << " = f" << id << ";\n";
  };
  auto is_pending = [&](uint32_t kind, uint32_t idx) { return kind == PROG_REG && !pending[idx].empty(); };
  auto value_of = [&](uint32_t kind, uint32_t idx) {  // an operand read as a VALUE
    if (is_pending(kind, idx)) flush(idx);
  };
  for (size_t i = 0; i < num_instrs; i++) {
    const uint32_t w0 = code[2 * i], w1 = code[2 * i + 1];
    const uint32_t op = w0 & 15u, dst = (w0 >> 4) & 15u, ak = (w0 >> 8) & 3u, bk = (w0 >> 10) & 3u;
    const uint32_t ai = w1 & 0xffffu, bi = w1 >> 16;
    need(ak, ai);
    if (op != PROG_NEG && op != PROG_MOV) need(bk, bi);
    if (fuse && op == PROG_MUL) {
      value_of(ak, ai), value_of(bk, bi);
      before_write(dst);
      // (dst may be one of the operands: the term then names the register's OLD value, which is what the variable holds
      //  until the register is materialised - nothing writes the variable before that: whoever overwrites a register
      //  first materialises every OTHER register whose terms name it, and its own pending value is either consumed or dead)
      pending[dst].assign(1, Term{false, ak, ai, bk, bi});
      continue;
    }
    if (fuse && (op == PROG_ADD || op == PROG_SUB) && is_pending(ak, ai) && is_pending(bk, bi) && ai != bi &&
        pending[ai].size() + pending[bi].size() <= JIT_DOT_MAX) {
      std::vector<Term> merged = pending[ai];
      for (Term t : pending[bi]) {
        if (op == PROG_SUB) t.neg = !t.neg;
        merged.push_back(t);
      }
      // (an operand register that is not the destination keeps its pending value: the expression compiler never reads a
      //  register twice, but nothing here relies on that)
      // (the new value is registered first: materialising the registers that still name the destination's old value writes
      //  THEIR variables, which the merged terms may name in turn - they must be visible to that bookkeeping)
      pending[dst] = merged;
      before_write(dst);
      continue;
    }
    if (fuse && op == PROG_NEG && is_pending(ak, ai)) {
      std::vector<Term> moved = pending[ai];
      for (Term& t : moved) t.neg = !t.neg;
      pending[dst] = moved;
      before_write(dst);
      continue;
    }
    // everything else reads values and writes where it stands
    value_of(ak, ai);
    if (op != PROG_NEG && op != PROG_MOV) value_of(bk, bi);
    before_write(dst);
    pending[dst].clear();
    if (op == PROG_NEG) {
      s << "      r" << dst << " = sub(Fr::zero(), " << operand(ak, ai) << ");\n";
    } else if (op == PROG_MOV) {
      s << "      r" << dst << " = " << operand(ak, ai) << ";\n";
    } else {
      const char* f = op == PROG_MUL ? "mul" : op == PROG_ADD ? "add" : "sub";
      s << "      r" << dst << " = " << f << "(" << operand(ak, ai) << ", " << operand(bk, bi) << ");\n";
    }
  }
  flush(result_reg);
  s << "      acc = add(acc, r" << result_reg
    << ");\n"
       "    }\n"
       "  }\n"
       // epilogue: wave sums to global memory, the workgroup that draws the launch's last ticket adds them up per
       // evaluation point and publishes to pinned memory + flag (the protocol of sc_round_prog_kernel)
       "  acc = wave_reduce_sum(acc);\n"
       // (with a lane buffer - dev.hpp ScFinishArgs::lanes, the protocol of resident.cuh fin_put / fin_get - the wave sums
       //  travel as self-validating 8-byte lanes and nobody fences)
       "  unsigned long long* const lanes = a.fin.lanes;\n"
       "  if (lane == 0) {\n"
       "    if (lanes) {\n"
       "      for (int k = 0; k < 8; k++)\n"
       "        __hip_atomic_store(&lanes[(group * D + wave) * 8 + k], (unsigned long long)acc.l[k] | ((unsigned long long)a.fin.seq << 32),\n"
       "                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n"
       "    } else {\n"
       "      a.partials[group * D + wave] = acc;\n"
       "      __threadfence();\n"
       "    }\n"
       "  }\n"
       "  __syncthreads();\n"
       "  if (threadIdx.x == 0) {\n"
       "    const unsigned t = lanes ? __hip_atomic_fetch_add(a.fin.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)\n"
       "                             : __hip_atomic_fetch_add(a.fin.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);\n"
       "    is_last = t == a.fin.last_ticket;\n"
       "  }\n"
       "  __syncthreads();\n"
       "  if (!is_last) return;\n"
       "  if (!lanes) __threadfence();\n"
       "  for (int x = w; x < D; x += NW) {\n"
       "    Fr a2 = Fr::zero();\n"
       "    for (unsigned long long i = lane; i < ngroups; i += 64) {\n"
       "      Fr p;\n"
       "      if (lanes) {\n"
       "        for (int k = 0; k < 8; k++) {\n"
       "          unsigned long long v = __hip_atomic_load(&lanes[(i * D + x) * 8 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n"
       "          for (unsigned spin = 0; (unsigned)(v >> 32) != a.fin.seq; spin++) {\n"
       "            if (spin > (1u << 22)) __builtin_trap();\n"
       "            __builtin_amdgcn_s_sleep(1);\n"
       "            v = __hip_atomic_load(&lanes[(i * D + x) * 8 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n"
       "          }\n"
       "          p.l[k] = (unsigned)v;\n"
       "        }\n"
       "      } else {\n"
       "        p = a.partials[i * D + x];\n"
       "      }\n"
       "      a2 = add(a2, p);\n"
       "    }\n"
       "    a2 = wave_reduce_sum(a2);\n"
       "    if (lane == 0) {\n"
       "      a.fin.out_host[x] = a2;\n"
       "      __threadfence_system();\n"
       "    }\n"
       "  }\n"
       "  __syncthreads();\n"
       // (dev.hpp ScFinishArgs::wide - sharded rounds of the all-reduce variant: the sums once more as tagged u64 lanes,
       //  the protocol of resident.cuh publish_round)
       "  if (threadIdx.x == 0) {\n"
       "    if (a.fin.wide)\n"
       "      for (int x = 0; x < D; x++)\n"
       "        for (int k = 0; k < 8; k++) {\n"
       "          const unsigned limb = __hip_atomic_load(&a.fin.out_host[x].l[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n"
       "          a.fin.wide[8 * x + k] = (unsigned long long)limb | ((unsigned long long)a.fin.tag << "
    << SC_LANE_TAG_SHIFT
    << ");\n"
       "        }\n"
       "    __hip_atomic_store(a.fin.flag, a.fin.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);\n"
       "  }\n"
       "}\n";
  return s.str();
}


}  // namespace
// development / tests: the source text the runtime compiler would be given for this program
std::string jit_debug_source(const uint32_t* code, size_t num_instrs, uint32_t num_regs, uint32_t result_reg, int degree) {
  return generate(code, num_instrs, num_regs, result_reg, degree);
}
namespace {
std::mutex g_mu;
// keyed on the whole program (device, register count, result register, degree, code words): no hash that could collide
std::map<std::vector<uint32_t>, JitKernel*> g_cache;  // never freed: modules live as long as the process
}  // namespace

// ---- code objects kept on disk: a process's first HyperPlonk proof otherwise pays seconds of hiprtc per distinct program
// (and every rank process of a node pays them again).  One file per (generated source, device headers, compiler version):
// LH_JIT_CACHE_DIR, else $XDG_CACHE_HOME/lasso_hip/jit, else $HOME/.cache/lasso_hip/jit; LH_JIT_CACHE=0 switches it off.
// A file is written under a temporary name and renamed; its header carries the payload's size and checksum, so a torn or
// foreign file is ignored and rewritten.
namespace {
uint64_t fnv1a(const void* p, size_t n, uint64_t h) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 0x100000001b3ull;
  return h;
}
struct DiskHeader {
  char magic[8];
  uint64_t size, sum;
  uint64_t src_hash[2];  // of the full generated source, the device headers and the compile options: compared on load
};
const char DISK_MAGIC[8] = {'L', 'H', 'J', 'I', 'T', '0', '2', 0};
// the compile options are part of what a cached code object was made from
const char* const JIT_OPTS[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
constexpr int JIT_NUM_OPTS = 3;
// only a directory / file that belongs to this user and that nobody else can write is trusted with code that decides proof
// bytes and writes device memory: anything else is ignored (the kernel is compiled, nothing is stored)
bool private_to_us(const struct stat& st) { return st.st_uid == geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0; }
bool make_dirs(const std::string& path) {  // mkdir -p
  for (size_t i = 1; i <= path.size(); i++)
    if (i == path.size() || path[i] == '/') {
      const std::string sub = path.substr(0, i);
      if (mkdir(sub.c_str(), 0700) != 0 && errno != EEXIST) return false;
    }
  return true;
}
const std::string& disk_dir() {  // empty: no disk cache
  static const std::string dir = [] {
    const char* on = getenv("LH_JIT_CACHE");
    if (on && atoi(on) == 0) return std::string();
    std::string d;
    if (const char* e = getenv("LH_JIT_CACHE_DIR")) d = e;
    else if (const char* x = getenv("XDG_CACHE_HOME")) d = std::string(x) + "/lasso_hip/jit";
    else if (const char* h = getenv("HOME")) d = std::string(h) + "/.cache/lasso_hip/jit";
    if (d.empty() || !make_dirs(d)) return std::string();
    struct stat st;
    if (stat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || !private_to_us(st)) {
      fprintf(stderr, "[lasso-hip] the JIT cache directory %s is not a private directory of this user: not used\n", d.c_str());
      return std::string();
    }
    return d;
  }();
  return dir;
}
void source_hash(const std::string& src, uint64_t out[2]);
std::string disk_name(const std::string& src) {
  static const uint64_t base[2] = {[] {
                                     int major = 0, minor = 0;
                                     (void)hiprtcVersion(&major, &minor);
                                     uint64_t h = fnv1a(&major, sizeof major, 0xcbf29ce484222325ull);
                                     h = fnv1a(&minor, sizeof minor, h);
                                     // (the runtime the code objects are made by and loaded into - hiprtc above and the HIP
                                     // runtime below, as they answer at RUN time - and the headers this library was built against)
                                     { int rt = 0; (void)hipRuntimeGetVersion(&rt); h = fnv1a(&rt, sizeof rt, h); }
                                     { int drv = 0; (void)hipDriverGetVersion(&drv); h = fnv1a(&drv, sizeof drv, h); }
                                     h = fnv1a(HIP_VERSION_GITHASH, sizeof HIP_VERSION_GITHASH, h);
                                     { const int patch = HIP_VERSION_PATCH; h = fnv1a(&patch, sizeof patch, h); }
                                     for (int i = 0; i < JIT_NUM_OPTS; i++) h = fnv1a(JIT_OPTS[i], strlen(JIT_OPTS[i]) + 1, h);
                                     h = fnv1a(JIT_FF_CUH, sizeof JIT_FF_CUH, h);
                                     h = fnv1a(JIT_FF_COLS_INC, sizeof JIT_FF_COLS_INC, h);
                                     return fnv1a(JIT_REDUCE_CUH, sizeof JIT_REDUCE_CUH, h);
                                   }(),
                                   0};
  const uint64_t h0 = fnv1a(src.data(), src.size(), base[0]);
  const uint64_t h1 = fnv1a(src.data(), src.size(), h0 ^ 0x9e3779b97f4a7c15ull);  // (a second, dependent pass: 128 bits of name)
  char buf[64];
  snprintf(buf, sizeof buf, "/gfx950-%016llx%016llx.co", (unsigned long long)h0, (unsigned long long)h1);
  return disk_dir() + buf;
}
// two independent 64-bit passes over everything a code object depends on (name and header use the same pair)
void source_hash(const std::string& src, uint64_t out[2]) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (int i = 0; i < JIT_NUM_OPTS; i++) h = fnv1a(JIT_OPTS[i], strlen(JIT_OPTS[i]) + 1, h);
  h = fnv1a(JIT_FF_CUH, sizeof JIT_FF_CUH, h);
  h = fnv1a(JIT_FF_COLS_INC, sizeof JIT_FF_COLS_INC, h);
  h = fnv1a(JIT_REDUCE_CUH, sizeof JIT_REDUCE_CUH, h);
  out[0] = fnv1a(src.data(), src.size(), h);
  out[1] = fnv1a(src.data(), src.size(), out[0] ^ 0x9e3779b97f4a7c15ull);
}
bool disk_load(const std::string& path, const std::string& src, std::string& bin) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  struct stat st;
  if (fstat(fileno(f), &st) != 0 || !S_ISREG(st.st_mode) || !private_to_us(st)) {  // (somebody else's file: not ours to run)
    fclose(f);
    return false;
  }
  DiskHeader hd;
  uint64_t want[2];
  source_hash(src, want);
  bool ok = fread(&hd, sizeof hd, 1, f) == 1 && memcmp(hd.magic, DISK_MAGIC, 8) == 0 && hd.size > 0 && hd.size < ((uint64_t)1 << 30) &&
            hd.src_hash[0] == want[0] && hd.src_hash[1] == want[1];
  if (ok) {
    bin.assign((size_t)hd.size, '\0');
    ok = fread(&bin[0], 1, bin.size(), f) == bin.size() && fnv1a(bin.data(), bin.size(), 0xcbf29ce484222325ull) == hd.sum;
  }
  fclose(f);
  return ok;
}
void disk_store(const std::string& path, const std::string& src, const std::string& bin) {
  char tmp[32];
  snprintf(tmp, sizeof tmp, ".tmp%ld", (long)getpid());
  const std::string t = path + tmp;
  const int fd = open(t.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
  FILE* f = fd >= 0 ? fdopen(fd, "wb") : nullptr;
  if (!f) {
    if (fd >= 0) close(fd);
    return;
  }
  DiskHeader hd;
  memcpy(hd.magic, DISK_MAGIC, 8);
  hd.size = bin.size(), hd.sum = fnv1a(bin.data(), bin.size(), 0xcbf29ce484222325ull);
  source_hash(src, hd.src_hash);
  const bool ok = fwrite(&hd, sizeof hd, 1, f) == 1 && fwrite(bin.data(), 1, bin.size(), f) == bin.size();
  if (fclose(f) != 0 || !ok || rename(t.c_str(), path.c_str()) != 0) (void)remove(t.c_str());
}
}  // namespace

bool jit_enabled(size_t num_vars) {
  static const int on = [] {
    const char* e = getenv("LH_EXPR_JIT");  // 0: always interpret
    return e ? atoi(e) : 1;
  }();
  static const size_t min_vars = [] {
    const char* e = getenv("LH_EXPR_JIT_MIN_VARS");  // smaller sum-checks are not worth seconds of compilation
    return e ? (size_t)atoll(e) : (size_t)16;
  }();
  return on && num_vars >= min_vars;
}

const JitKernel* jit_sc_round(const Ctx& c, const uint32_t* code, size_t num_instrs, uint32_t num_regs, uint32_t result_reg,
                              int degree) {
  // a module belongs to the device it was loaded on: the ctx's (made current by the C-ABI entry point)
  std::vector<uint32_t> key{(uint32_t)c.device, num_regs, result_reg, (uint32_t)degree};
  key.insert(key.end(), code, code + 2 * num_instrs);
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_cache.find(key);
  if (it != g_cache.end()) return it->second->failed ? nullptr : it->second;
  JitKernel* k = new JitKernel();
  g_cache[key] = k;
  k->failed = true;
  const auto t0 = std::chrono::steady_clock::now();
  const std::string src = generate(code, num_instrs, num_regs, result_reg, degree);
  std::string bin;
  const std::string disk = disk_dir().empty() ? std::string() : disk_name(src);
  bool from_disk = !disk.empty() && disk_load(disk, src, bin);
  if (from_disk && hipModuleLoadData(&k->mod, bin.data()) != hipSuccess) {  // (unloadable: compile it again and overwrite)
    (void)hipGetLastError();
    k->mod = nullptr;
    from_disk = false;
  }
  size_t n = bin.size();
  if (!from_disk) {
    hiprtcProgram prog;
    const char* hdr[] = {JIT_FF_CUH, JIT_REDUCE_CUH, JIT_FF_COLS_INC};
    const char* names[] = {"ff.cuh", "reduce.cuh", "ff_cols.inc"};
    if (hiprtcCreateProgram(&prog, src.c_str(), "sc_round_jit.hip", 3, hdr, names) != HIPRTC_SUCCESS) return nullptr;
    const hiprtcResult r = hiprtcCompileProgram(prog, JIT_NUM_OPTS, (const char**)JIT_OPTS);
    if (r != HIPRTC_SUCCESS) {
      size_t ln = 0;
      (void)hiprtcGetProgramLogSize(prog, &ln);
      std::string log(ln + 1, '\0');
      (void)hiprtcGetProgramLog(prog, &log[0]);
      fprintf(stderr, "[lasso-hip] runtime compilation of a sum-check program failed (%s), interpreting it instead:\n%.2000s\n",
              hiprtcGetErrorString(r), log.c_str());
      (void)hiprtcDestroyProgram(&prog);
      return nullptr;
    }
    (void)hiprtcGetCodeSize(prog, &n);
    bin.assign(n, '\0');
    (void)hiprtcGetCode(prog, &bin[0]);
    (void)hiprtcDestroyProgram(&prog);
    if (hipModuleLoadData(&k->mod, bin.data()) != hipSuccess) return nullptr;
    if (!disk.empty()) disk_store(disk, src, bin);
  }
  if (hipModuleGetFunction(&k->fn, k->mod, "sc_round_jit") != hipSuccess) return nullptr;
  k->threads = 256u;
  k->degree = (unsigned)degree;
  k->failed = false;
  int per_cu = 0;
  if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k->fn, (int)k->threads, 0) == hipSuccess && per_cu > 0)
    k->blocks_per_cu = (unsigned)per_cu;
  if (getenv("LH_HP_DEBUG")) {
    int vgprs = 0, scratch = 0;
    (void)hipFuncGetAttribute(&vgprs, HIP_FUNC_ATTRIBUTE_NUM_REGS, k->fn);
    (void)hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, k->fn);
    fprintf(stderr, "[expr] %s a %zu-instruction program for degree %d in %.2f s (%zu B of code, %d registers, %d B scratch, %u workgroups per CU)\n",
            from_disk ? "loaded from the disk cache" : "compiled", num_instrs, degree, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), n, vgprs, scratch,
            k->blocks_per_cu);
  }
  return k;
}

unsigned jit_blocks_per_cu(const JitKernel* k) { return k->blocks_per_cu; }

// (`points`: the launch evaluates at X = 1..points - the kernel reads it off its grid: the expression's degree, or fewer in
//  an eq-factored round)
void jit_launch(Ctx& c, const JitKernel* k, const ProgRound& pr, unsigned points, unsigned grid, size_t size, Fr* partials,
                const ScFinishArgs& fin) {
  JitArgs a;
  for (int i = 0; i < SC_MAX_TABLES; i++) a.in[i] = pr.in[i];
  a.consts = pr.consts;
  a.size = size;
  a.partials = partials;
  a.fin = fin;
  size_t bytes = sizeof(a);
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &bytes, HIP_LAUNCH_PARAM_END};
  LH_HIP(hipModuleLaunchKernel(k->fn, points, grid, 1, k->threads, 1, 1, 0, c.stream, nullptr, extra));
}

}  // namespace lh
