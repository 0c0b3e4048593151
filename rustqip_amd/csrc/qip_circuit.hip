// qip_circuit.hip — apply_ops: gate fusion, tile sweeps (interpreter launches and run-time-compiled segments), hipGraph programs.
#include "qip_tile.h"
#include <atomic>
#include <thread>

#include <mutex>

#include <cerrno>
#include <climits>
#include <dirent.h>
#include <fcntl.h>
#include <sched.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
extern char** environ;

// ---------------------------------------------------------------------------------------
// gate fusion: option "fuse" = K merges consecutive small gates into dense
// gates on <= K qubits, applied in ONE sweep each.  The reference has no analogue (its apply_ops
// multi-op path is unused and inconsistent); this is pure host bookkeeping on
// 2^K x 2^K matrices.  Results equal the gate-by-gate path up to the rounding of the matrix
// products (|delta| ~ 1e-15 per fused gate), so fused runs are held to the 1e-12 bar, never to
// bit equality.  Open clusters always act on pairwise disjoint qubit sets, so they commute and
// may be flushed in any order; an op is merged only into clusters it overlaps, or into a
// disjoint one (which also commutes with every other open cluster).
// ---------------------------------------------------------------------------------------
typedef std::complex<double> cd;
struct Cluster {
  std::vector<uint32_t> pos;  // bit positions, descending: pos[0] is the MSB of the sub-index
  std::vector<cd> m;          // 2^q x 2^q, row-major
};

// dense matrix of a flattened op over its own index list (controls first), MSB-first order
template <typename T>
static bool op_to_dense(uint32_t n, const FlatOp& f, uint32_t max_k, std::vector<uint32_t>* pos,
                        std::vector<cd>* mat) {
  if (!f.distinct || f.k_all > max_k) return false;
  const uint32_t kt = f.k_all, k = f.n_op;
  const size_t S = (size_t)1 << kt, Si = (size_t)1 << k, thr = S - Si;
  pos->clear();
  for (uint32_t j = 0; j < kt; ++j) pos->push_back(n - 1 - (uint32_t)f.outer->indices[j]);
  mat->assign(S * S, cd(0, 0));
  for (size_t r = 0; r < thr; ++r) (*mat)[r * S + r] = cd(1, 0);
  const T* dense = static_cast<const T*>(f.inner->dense);
  const T* vals = static_cast<const T*>(f.inner->sparse_vals);
  for (size_t r = 0; r < Si; ++r) {
    switch (f.inner->kind) {
      case QIP_OP_MATRIX:
        for (size_t c = 0; c < Si; ++c)
          (*mat)[(thr + r) * S + thr + c] = cd(dense[2 * (r * Si + c)], dense[2 * (r * Si + c) + 1]);
        break;
      case QIP_OP_SPARSE:
        for (uint64_t e = f.inner->sparse_rowptr[r]; e < f.inner->sparse_rowptr[r + 1]; ++e)
          (*mat)[(thr + r) * S + thr + f.inner->sparse_cols[e]] += cd(vals[2 * e], vals[2 * e + 1]);
        break;
      default: {  // SWAP
        const uint32_t h = k >> 1;
        const size_t c = ((r & (((size_t)1 << h) - 1)) << h) + (r >> h);
        (*mat)[(thr + r) * S + thr + c] = cd(1, 0);
      }
    }
  }
  return true;
}

// matrix of `m` (over positions `pos`, MSB first) embedded into the space of `upos` (descending)
static std::vector<cd> embed(const std::vector<cd>& m, const std::vector<uint32_t>& pos,
                             const std::vector<uint32_t>& upos) {
  const uint32_t u = (uint32_t)upos.size(), k = (uint32_t)pos.size();
  const size_t U = (size_t)1 << u, S = (size_t)1 << k;
  std::vector<uint32_t> bit_in_u(k);
  size_t opmask = 0;
  for (uint32_t i = 0; i < k; ++i)
    for (uint32_t j = 0; j < u; ++j)
      if (upos[j] == pos[i]) {
        bit_in_u[i] = u - 1 - j;
        opmask |= (size_t)1 << (u - 1 - j);
      }
  auto sub = [&](size_t r) {
    size_t s_ = 0;
    for (uint32_t i = 0; i < k; ++i) s_ |= ((r >> bit_in_u[i]) & 1) << (k - 1 - i);
    return s_;
  };
  std::vector<cd> e(U * U, cd(0, 0));
  for (size_t r = 0; r < U; ++r)
    for (size_t c = 0; c < U; ++c)
      if ((r & ~opmask) == (c & ~opmask)) e[r * U + c] = m[sub(r) * S + sub(c)];
  return e;
}

static void cluster_apply(Cluster* cl, const std::vector<uint32_t>& pos, const std::vector<cd>& m) {
  std::vector<uint32_t> upos = cl->pos;
  for (uint32_t p : pos)
    if (std::find(upos.begin(), upos.end(), p) == upos.end()) upos.push_back(p);
  std::sort(upos.begin(), upos.end(), std::greater<uint32_t>());
  const size_t U = (size_t)1 << upos.size();
  const std::vector<cd> a = embed(m, pos, upos);              // the new gate, applied after
  const std::vector<cd> b = embed(cl->m, cl->pos, upos);      // what the cluster already holds
  std::vector<cd> out(U * U, cd(0, 0));
  for (size_t r = 0; r < U; ++r)
    for (size_t k = 0; k < U; ++k) {
      const cd v = a[r * U + k];
      if (v == cd(0, 0)) continue;
      for (size_t c = 0; c < U; ++c) out[r * U + c] += v * b[k * U + c];
    }
  cl->pos = upos;
  cl->m = out;
}

template <typename T>
static int flush_cluster(qip_hip_state* s, const Cluster& cl) {
  const size_t S = (size_t)1 << cl.pos.size();
  std::vector<uint64_t> idx;
  for (uint32_t p : cl.pos) idx.push_back(s->n - 1 - p);
  std::vector<T> data(2 * S * S);
  for (size_t e = 0; e < S * S; ++e) {
    data[2 * e] = (T)cl.m[e].real();
    data[2 * e + 1] = (T)cl.m[e].imag();
  }
  qip_op op;
  memset(&op, 0, sizeof op);
  op.kind = QIP_OP_MATRIX;
  op.n_indices = (uint32_t)idx.size();
  op.indices = idx.data();
  op.dense = data.data();
  return apply_op_t<T>(s, &op);
}

template <typename T>
int apply_ops_fused(qip_hip_state* s, const qip_op* ops, uint64_t count, uint32_t K) {
  std::vector<Cluster> open;
  auto overlaps = [](const Cluster& c, const std::vector<uint32_t>& pos) {
    for (uint32_t p : pos)
      if (std::find(c.pos.begin(), c.pos.end(), p) != c.pos.end()) return true;
    return false;
  };
  auto flush_overlapping = [&](const std::vector<uint32_t>& pos) -> int {
    for (size_t i = 0; i < open.size();) {
      if (overlaps(open[i], pos)) {
        QCHK(flush_cluster<T>(s, open[i]));
        open.erase(open.begin() + i);
      } else {
        ++i;
      }
    }
    return QIP_OK;
  };
  for (uint64_t i = 0; i < count; ++i) {
    FlatOp f;
    int rc = flatten_op(s->n, &ops[i], false, &f);
    if (rc != QIP_OK) {
      std::string msg = g_last_error;
      return fail(rc, "op %llu: %s", (unsigned long long)i, msg.c_str());
    }
    std::vector<uint32_t> pos;
    std::vector<cd> m;
    if (!op_to_dense<T>(s->n, f, K, &pos, &m)) {
      // not fusable (too many qubits / repeated indices): everything it touches goes first
      std::vector<uint32_t> all;
      for (uint32_t j = 0; j < f.k_all; ++j) all.push_back(s->n - 1 - (uint32_t)f.outer->indices[j]);
      QCHK(flush_overlapping(all));
      QCHK(apply_op_t<T>(s, &ops[i]));
      continue;
    }
    std::vector<size_t> hit;
    size_t union_size = pos.size();
    for (size_t c = 0; c < open.size(); ++c)
      if (overlaps(open[c], pos)) {
        hit.push_back(c);
        for (uint32_t p : open[c].pos)
          if (std::find(pos.begin(), pos.end(), p) == pos.end()) ++union_size;
      }
    if (hit.empty()) {
      // disjoint from every open cluster: join the fullest one that still has room
      size_t best = open.size();
      for (size_t c = 0; c < open.size(); ++c)
        if (open[c].pos.size() + pos.size() <= K && (best == open.size() || open[c].pos.size() > open[best].pos.size()))
          best = c;
      if (best == open.size()) {
        Cluster cl;
        cl.pos = {};
        cl.m = {cd(1, 0)};
        open.push_back(cl);
      }
      cluster_apply(&open[best], pos, m);
    } else if (union_size <= K) {
      // merge the overlapped clusters (mutually disjoint => their product is a Kronecker product)
      for (size_t h = 1; h < hit.size(); ++h) cluster_apply(&open[hit[0]], open[hit[h]].pos, open[hit[h]].m);
      for (size_t h = hit.size(); h-- > 1;) open.erase(open.begin() + hit[h]);
      cluster_apply(&open[hit[0]], pos, m);
    } else {
      QCHK(flush_overlapping(pos));
      Cluster cl;
      cl.pos = {};
      cl.m = {cd(1, 0)};
      cluster_apply(&cl, pos, m);
      open.push_back(cl);
    }
  }
  for (const Cluster& cl : open) QCHK(flush_cluster<T>(s, cl));
  return QIP_OK;
}

// ---------------------------------------------------------------------------------------
// Segment-specialised tile sweeps (option "tile_jit"): the interpreter k_tile_passes spends most of its issue slots
// on decoding — per gate and per wave ~54 scalar + ~58 vector instructions that only depend on the segment
// (profiles/r01_tile_pmc.md).  Here the host writes the segment out as straight-line HIP source — the same load /
// pass / store skeleton, one call of the very same pass_* helper per gate with the gate descriptor as a constexpr
// value — and compiles it with hiprtc against the embedded qip_kernels.h.  Every op code, bit position, control mask,
// zero / real / X flag folds away; what is left per gate is its arithmetic, operation for operation what the
// interpreter executes, so results are bit-identical.  Kernels are cached per process by their source text, so a
// circuit replayed many times (programs, variational loops with fixed angles) compiles once (~0.3-1 s per segment).
// ---------------------------------------------------------------------------------------
static const char kKernelsHeaderSrc[] =
#include "qip_kernels_embed.inc"
    ;

struct Hiprtc {
  void* handle = nullptr;
  int (*CreateProgram)(void**, const char*, const char*, int, const char**, const char**) = nullptr;
  int (*CompileProgram)(void*, int, const char**) = nullptr;
  int (*GetProgramLogSize)(void*, size_t*) = nullptr;
  int (*GetProgramLog)(void*, char*) = nullptr;
  int (*GetCodeSize)(void*, size_t*) = nullptr;
  int (*GetCode)(void*, char*) = nullptr;
  int (*DestroyProgram)(void**) = nullptr;
};
static Hiprtc g_rtc;
// The run-time compiler state below (loader, kernel cache, counters) is process-global while handles are per thread —
// "separate handles are independent" (include/qip_hip.h) has to hold for two threads that both use tile_jit: one mutex
// serialises loading, the cache and the compilation itself (hiprtc calls are not documented to be re-entrant).
static std::mutex g_jit_mutex;
static int hiprtc_load() {
  if (g_rtc.handle) return QIP_OK;
  void* h = nullptr;
  for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
    h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) return fail(QIP_ERR_UNSUPPORTED, "option tile_jit needs libhiprtc: %s", dlerror());
#define RSYM(field, name)                                                           \
  do {                                                                              \
    *(void**)(&g_rtc.field) = dlsym(h, name);                                       \
    if (!g_rtc.field) return fail(QIP_ERR_UNSUPPORTED, "libhiprtc lacks %s", name); \
  } while (0)
  RSYM(CreateProgram, "hiprtcCreateProgram");
  RSYM(CompileProgram, "hiprtcCompileProgram");
  RSYM(GetProgramLogSize, "hiprtcGetProgramLogSize");
  RSYM(GetProgramLog, "hiprtcGetProgramLog");
  RSYM(GetCodeSize, "hiprtcGetCodeSize");
  RSYM(GetCode, "hiprtcGetCode");
  RSYM(DestroyProgram, "hiprtcDestroyProgram");
#undef RSYM
  g_rtc.handle = h;
  return QIP_OK;
}

// source -> code object (host only: works without a device, which is how the CPU tests cover it)
static int hiprtc_compile(const std::string& src, bool fma, std::vector<char>* code) {
  QCHK(hiprtc_load());
  void* prog = nullptr;
  const char* hdr_src[1] = {kKernelsHeaderSrc};
  const char* hdr_name[1] = {"qip_kernels.h"};
  if (g_rtc.CreateProgram(&prog, src.c_str(), "qip_segment.hip", 1, hdr_src, hdr_name) != 0)
    return fail(QIP_ERR_DEVICE, "hiprtcCreateProgram failed");
  // as rustqip_amd/build.py; `fma` (option "tile_fma", only honoured for tile = 2, whose bar is 1e-12 anyway): products may
  // fuse into the sums that consume them (v_fma_f64): a complex product is 4 instead of 6 vector instructions
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", fma ? "-ffp-contract=fast" : "-ffp-contract=off", "-fno-slp-vectorize"};
  const int rc = g_rtc.CompileProgram(prog, (int)(sizeof opts / sizeof opts[0]), opts);
  if (rc != 0) {
    size_t n = 0;
    g_rtc.GetProgramLogSize(prog, &n);
    std::string log(n + 1, '\0');
    if (n) g_rtc.GetProgramLog(prog, &log[0]);
    g_rtc.DestroyProgram(&prog);
    return fail(QIP_ERR_DEVICE, "hiprtc could not compile a tile segment (%d): %.800s", rc, log.c_str());
  }
  size_t sz = 0;
  g_rtc.GetCodeSize(prog, &sz);
  code->resize(sz);
  g_rtc.GetCode(prog, code->data());
  g_rtc.DestroyProgram(&prog);
  return QIP_OK;
}


// ---------------------------------------------------------------------------------------
// r5: code objects on disk, and a plan's new segments compiled in helper PROCESSES.
//  * hiprtc costs ~0.45 s per 11-bit segment and ~1.3 s per wide one, and the code-object manager behind it serialises the
//    compilations of one process (r4: 8 host threads bought nothing).  Separate processes do scale (measured on 8 cores: four
//    processes compile four plans in the time of one), so the segments of a plan that are new are written out as source files,
//    `qip_jitc` (a 30-line host program next to the library; it calls qip_hip_jit_compile_file below) is spawned up to
//    "jit_procs" times, and the code objects come back through the disk cache.  No helper, one new segment, or jit_procs = 1:
//    the compilation happens in this process as before.
//  * every code object is kept under $QIP_HIP_CACHE_DIR (default $XDG_CACHE_HOME/qip_hip or ~/.cache/qip_hip; "off" or an
//    empty value disables), named by a 128-bit hash of compiler version + flags + the embedded kernel header + the source text,
//    written to a temporary name and renamed (readers never see half a file; two processes that compile the same segment just
//    both succeed).  A second process loads instead of compiling: ~1 ms per segment.
//  * r6, what is TRUSTED: a code object found on disk ends up on the GPU, so the cache only ever reads what this user wrote.
//    The directory must be owned by the effective uid and writable by nobody else (an existing directory that is not: no disk
//    cache — the default rule falls back to "none", qip_hip_jit_set_cache_dir reports the reason); a file must be a regular
//    file (no symlink), owned by the uid, not group/world-writable; its header carries the source length, the second word of
//    the key and a hash of the CODE BYTES, all checked on load.  Anything odd is a miss (and a recompilation that overwrites
//    the file), never an error, never a load.  The key names the compiler by the ROCm installation the helper processes use
//    ($ROCM_PATH or /opt/rocm: its release and the sizes of its libhiprtc / libamd_comgr — see jit_compiler_identity; a host
//    program that has another ROCm's libraries loaded, PyTorch for one, hands even single segments to a helper).
//  * r6, bounded: after a store the directory is trimmed to "jit_disk_cap_mb" (global option; $QIP_HIP_CACHE_MAX_MB; default
//    1024) — oldest modification time first (a hit refreshes it), stale temporaries with them.
// ---------------------------------------------------------------------------------------
int64_t g_jit_disk = 1;   // global option "jit_disk_cache"
int64_t g_jit_procs = 0;  // global option "jit_procs": 0 = automatic (the CPUs this process may use, at most 16), 1 = in process only
int64_t g_jit_world = 1;  // ranks that share this host's CPUs (set by qip_hip_dist_create): automatic jit_procs is divided by it
static std::string g_jit_dir;           // resolved cache directory ("" = none)
static bool g_jit_dir_resolved = false;
static uint64_t g_jit_disk_hits = 0, g_jit_disk_stores = 0, g_jit_helper_procs = 0, g_jit_helper_segments = 0;
static double g_jit_load_ms = 0;
static uint64_t g_jit_background_segments = 0;  // segments handed to background helpers (option tile_auto, one-shot callers)

struct JitHash {
  uint64_t a = 0, b = 0;
};
static uint64_t hash_fnv(const char* p, size_t n, uint64_t h) {
  for (size_t i = 0; i < n; ++i) {
    h ^= (unsigned char)p[i];
    h *= 1099511628211ull;
  }
  return h;
}
static uint64_t hash_mix(const char* p, size_t n, uint64_t h) {  // an independent second word: 8 bytes at a time
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    memcpy(&w, p + i, 8);
    h = (h ^ w) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
  }
  for (; i < n; ++i) {
    h = (h ^ (unsigned char)p[i]) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 31;
  }
  return h;
}
static const char* jit_flags_text(bool fma) {
  return fma ? "gfx950 -O3 c++17 contract=fast no-slp" : "gfx950 -O3 c++17 contract=off no-slp";
}
// The compiler a key names.  New segments are compiled by helper processes (qip_jitc): fresh processes that load the system's
// ROCm — whatever THIS process has loaded.  A host program that imported PyTorch first carries PyTorch's own bundled libhiprtc /
// libamd_comgr / HIP runtime (another ROCm release), so "as loaded here" named a different compiler than the one that wrote the
// files: the helpers' objects failed the header check and every segment was compiled a second time in process (r6: bench.py's
// 207 segments, 93 s instead of 31), and a process without PyTorch never found what a process with PyTorch had cached.
// So the identity is that of the INSTALLATION the helpers use, read from its files: $ROCM_PATH or /opt/rocm — the release
// (.info/version) and the byte sizes of libhiprtc / libamd_comgr there (an updated ROCm changes at least one of the three).
// Only when no such installation is found does the key fall back to what this process has loaded (hiprtc's version, the HIP
// runtime's version, the sizes of the two shared objects as loaded).
static std::string g_jit_canonical_hiprtc;  // real path of the installation's libhiprtc ("" = fell back to as-loaded)
static std::string jit_compiler_identity() {
  for (const char* root : {(const char*)getenv("ROCM_PATH"), "/opt/rocm"}) {
    if (!root || !*root) continue;
    const std::string r = root;
    struct stat st_rtc, st_comgr;
    if (stat((r + "/lib/libhiprtc.so").c_str(), &st_rtc) != 0 || stat((r + "/lib/libamd_comgr.so").c_str(), &st_comgr) != 0) continue;
    char release[64] = "?";
    if (FILE* f = fopen((r + "/.info/version").c_str(), "r")) {
      if (fscanf(f, "%63s", release) != 1) strcpy(release, "?");
      fclose(f);
    }
    char real[PATH_MAX];
    if (realpath((r + "/lib/libhiprtc.so").c_str(), real)) g_jit_canonical_hiprtc = real;
    char v[224];
    snprintf(v, sizeof v, "qipjit3 rocm %s libhiprtc %lld comgr %lld ", release, (long long)st_rtc.st_size, (long long)st_comgr.st_size);
    return v;
  }
  int major = 0, minor = 0, rt = 0;
  long long sz_rtc = 0, sz_comgr = 0;
  (void)hiprtc_load();  // (part of every key: always asked, so that keys do not depend on who hashes first)
  if (g_rtc.handle) {
    int (*ver)(int*, int*) = nullptr;
    *(void**)(&ver) = dlsym(g_rtc.handle, "hiprtcVersion");
    if (ver) (void)ver(&major, &minor);
    Dl_info info;
    struct stat st;
    if (ver && dladdr((void*)ver, &info) && info.dli_fname && stat(info.dli_fname, &st) == 0) sz_rtc = (long long)st.st_size;
    // (loaded here if hiprtc has not done so yet — it does lazily, at its first compilation: the key must not depend on that)
    for (const char* name : {"libamd_comgr.so", "libamd_comgr.so.3", "/opt/rocm/lib/libamd_comgr.so"}) {
      void* comgr = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (!comgr) continue;
      void* sym = dlsym(comgr, "amd_comgr_get_version");
      if (sym && dladdr(sym, &info) && info.dli_fname && stat(info.dli_fname, &st) == 0) sz_comgr = (long long)st.st_size;
      break;  // (stays loaded: hiprtc needs it anyway)
    }
  }
  (void)hipRuntimeGetVersion(&rt);
  char v[160];
  snprintf(v, sizeof v, "qipjit2 hiprtc %d.%d runtime %d libhiprtc %lld comgr %lld ", major, minor, rt, sz_rtc, sz_comgr);
  return v;
}
// true when a compilation in THIS process would use another libhiprtc than the one the keys name (a host program that brought its
// own ROCm libraries): such a process sends even a single new segment to a helper, so that a file is what its name says
static bool jit_foreign_hiprtc_loaded() {
  static const bool foreign = [] {
    if (g_jit_canonical_hiprtc.empty() || hiprtc_load() != QIP_OK) return false;
    int (*ver)(int*, int*) = nullptr;
    *(void**)(&ver) = dlsym(g_rtc.handle, "hiprtcVersion");
    Dl_info info;
    char real[PATH_MAX];
    if (!ver || !dladdr((void*)ver, &info) || !info.dli_fname || !realpath(info.dli_fname, real)) return false;
    return g_jit_canonical_hiprtc != real;
  }();
  return foreign;
}
static JitHash jit_hash(const std::string& src, bool fma) {
  static JitHash base = [] {  // the embedded header and the compiler's identity: once per process
    JitHash h;
    const std::string v = jit_compiler_identity();
    h.a = hash_fnv(v.data(), v.size(), 1469598103934665603ull);
    h.b = hash_mix(v.data(), v.size(), 0x243F6A8885A308D3ull);
    h.a = hash_fnv(kKernelsHeaderSrc, sizeof kKernelsHeaderSrc, h.a);
    h.b = hash_mix(kKernelsHeaderSrc, sizeof kKernelsHeaderSrc, h.b);
    return h;
  }();
  JitHash h = base;
  const char* f = jit_flags_text(fma);
  h.a = hash_fnv(f, strlen(f), h.a);
  h.b = hash_mix(f, strlen(f), h.b);
  h.a = hash_fnv(src.data(), src.size(), h.a);
  h.b = hash_mix(src.data(), src.size(), h.b);
  return h;
}
static bool mkdir_p(const std::string& dir) {
  struct stat st;
  if (stat(dir.c_str(), &st) == 0) return S_ISDIR(st.st_mode);
  const size_t slash = dir.find_last_of('/');
  if (slash != std::string::npos && slash > 0 && !mkdir_p(dir.substr(0, slash))) return false;
  return mkdir(dir.c_str(), 0700) == 0 || (stat(dir.c_str(), &st) == 0 && S_ISDIR(st.st_mode));
}
// A cache directory is used only when it belongs to this user and nobody else can write to it: what it holds is loaded onto the GPU.
static bool jit_dir_trusted(const std::string& dir, std::string* why) {
  struct stat st;
  if (stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) {
    if (why) *why = "not a directory";
    return false;
  }
  if (st.st_uid != geteuid()) {
    if (why) *why = "not owned by this user";
    return false;
  }
  if (st.st_mode & (S_IWGRP | S_IWOTH)) {
    if (why) *why = "writable by group or others";
    return false;
  }
  return access(dir.c_str(), W_OK | X_OK) == 0 || ((why ? (void)(*why = "not writable") : (void)0), false);
}
// under g_jit_mutex (or before any thread exists): where code objects are kept; "" = nowhere
static const std::string& jit_dir_locked() {
  if (g_jit_dir_resolved) return g_jit_dir;
  g_jit_dir_resolved = true;
  g_jit_dir.clear();
  std::string dir;
  if (const char* e = getenv("QIP_HIP_CACHE_DIR")) {
    if (!*e || !strcmp(e, "off") || !strcmp(e, "0")) return g_jit_dir;
    dir = e;
  } else if (const char* x = getenv("XDG_CACHE_HOME")) {
    if (*x) dir = std::string(x) + "/qip_hip";
  }
  if (dir.empty()) {
    const char* home = getenv("HOME");
    if (!home || !*home) return g_jit_dir;
    dir = std::string(home) + "/.cache/qip_hip";
  }
  if (mkdir_p(dir) && jit_dir_trusted(dir, nullptr)) g_jit_dir = dir;
  return g_jit_dir;
}
static std::string jit_disk_path(const std::string& dir, const JitHash& h) {
  char name[48];
  snprintf(name, sizeof name, "/%016llx%016llx.co", (unsigned long long)h.a, (unsigned long long)h.b);
  return dir + name;
}
struct JitFileHeader {
  char magic[8];
  uint64_t src_len, hash_b, code_len, code_hash;
};
static uint64_t jit_code_hash(const char* p, size_t n) { return hash_mix(p, n, 0x13198A2E03707344ull ^ (uint64_t)n); }
static bool jit_disk_read(const std::string& path, size_t src_len, const JitHash& h, std::vector<char>* code) {
  const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
  if (fd < 0) return false;
  struct stat st;
  bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == geteuid() && !(st.st_mode & (S_IWGRP | S_IWOTH));
  JitFileHeader hd;
  auto read_all = [&](void* dst, size_t n) {
    char* q = (char*)dst;
    while (n) {
      const ssize_t got = read(fd, q, n);
      if (got < 0 && errno == EINTR) continue;
      if (got <= 0) return false;
      q += got;
      n -= (size_t)got;
    }
    return true;
  };
  ok = ok && read_all(&hd, sizeof hd) && !memcmp(hd.magic, "QIPJIT2", 8) && hd.src_len == src_len && hd.hash_b == h.b && hd.code_len > 0 &&
       hd.code_len < (1ull << 31) && (uint64_t)st.st_size == sizeof hd + hd.code_len;
  if (ok) {
    code->resize(hd.code_len);
    ok = read_all(code->data(), hd.code_len) && jit_code_hash(code->data(), code->size()) == hd.code_hash;
  }
  if (ok) (void)futimens(fd, nullptr);  // a hit refreshes the modification time: the trim below drops the oldest first
  close(fd);
  if (!ok) code->clear();
  return ok;
}
static std::atomic<unsigned> g_jit_tmp_serial{0};  // temporaries of one process never share a name (two handles may compile side by side)
static int64_t g_jit_disk_cap_mb = -1;             // global option "jit_disk_cap_mb"; -1 = $QIP_HIP_CACHE_MAX_MB or 1024
static uint64_t g_jit_disk_trimmed = 0;
static bool jit_disk_write(const std::string& path, size_t src_len, const JitHash& h, const std::vector<char>& code) {
  char suffix[64];
  snprintf(suffix, sizeof suffix, ".tmp.%ld.%u", (long)getpid(), g_jit_tmp_serial.fetch_add(1));
  const std::string tmp = path + suffix;
  const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
  if (fd < 0) return false;
  JitFileHeader hd;
  memcpy(hd.magic, "QIPJIT2", 8);
  hd.src_len = src_len;
  hd.hash_b = h.b;
  hd.code_len = code.size();
  hd.code_hash = jit_code_hash(code.data(), code.size());
  auto write_all = [&](const void* src, size_t n) {
    const char* q = (const char*)src;
    while (n) {
      const ssize_t put = write(fd, q, n);
      if (put < 0 && errno == EINTR) continue;
      if (put <= 0) return false;
      q += put;
      n -= (size_t)put;
    }
    return true;
  };
  bool ok = write_all(&hd, sizeof hd) && write_all(code.data(), code.size());
  ok = close(fd) == 0 && ok;
  if (ok) ok = rename(tmp.c_str(), path.c_str()) == 0;
  if (!ok) (void)unlink(tmp.c_str());
  return ok;
}
// Keep the directory under its bound: code objects by modification time, oldest first, until 90 % of the bound is left; and the
// temporaries (sources handed to helpers, half-written objects) that a killed process left behind more than an hour ago.
static void jit_disk_trim(const std::string& dir) {
  int64_t cap_mb = g_jit_disk_cap_mb;
  if (cap_mb < 0) {
    const char* e = getenv("QIP_HIP_CACHE_MAX_MB");
    cap_mb = e && *e ? atoll(e) : 1024;
  }
  if (cap_mb <= 0 || dir.empty()) return;
  DIR* d = opendir(dir.c_str());
  if (!d) return;
  struct Ent {
    std::string name;
    int64_t mtime_ns;
    uint64_t size;
  };
  std::vector<Ent> objs;
  uint64_t total = 0;
  const time_t now = time(nullptr);
  while (struct dirent* de = readdir(d)) {
    const std::string name = de->d_name;
    struct stat st;
    if (name == "." || name == ".." || lstat((dir + "/" + name).c_str(), &st) != 0 || !S_ISREG(st.st_mode)) continue;
    const bool obj = name.size() > 3 && name.compare(name.size() - 3, 3, ".co") == 0;
    if (obj) {
      objs.push_back({name, (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec, (uint64_t)st.st_size});
      total += (uint64_t)st.st_size;
    } else if ((name.find(".co.tmp.") != std::string::npos || name.compare(0, 4, "seg.") == 0) && now - st.st_mtime > 3600) {
      (void)unlink((dir + "/" + name).c_str());
    }
  }
  closedir(d);
  const uint64_t cap = (uint64_t)cap_mb << 20;
  if (total <= cap) return;
  std::sort(objs.begin(), objs.end(), [](const Ent& a, const Ent& b) { return a.mtime_ns < b.mtime_ns; });
  for (const Ent& e : objs) {
    if (total <= cap / 10 * 9) break;
    if (unlink((dir + "/" + e.name).c_str()) == 0) {
      total -= e.size;
      g_jit_disk_trimmed += 1;
    }
  }
}
int jit_set_disk_cap_mb(int64_t mb) {
  std::lock_guard<std::mutex> lock(g_jit_mutex);
  g_jit_disk_cap_mb = mb;
  return QIP_OK;
}

// Host only (no device): compile the segment source in `src_path` and leave the code object at `out_path` in the disk
// cache's file format.  This is what the helper processes run (tools/qip_jitc.c).
extern "C" int qip_hip_jit_compile_file(const char* src_path, int fma, const char* out_path) try {
  if (!src_path || !out_path) return fail(QIP_ERR_INVALID, "null path");
  FILE* f = fopen(src_path, "rb");
  if (!f) return fail(QIP_ERR_INVALID, "cannot read %s", src_path);
  std::string src;
  char buf[1 << 16];
  size_t got;
  while ((got = fread(buf, 1, sizeof buf, f)) > 0) src.append(buf, got);
  fclose(f);
  std::vector<char> code;
  QCHK(hiprtc_compile(src, fma != 0, &code));
  // The requester named the file by ITS key (<32 hex digits>.co): the header repeats the key's second word so that the requester's
  // own check passes — the name is the contract, whatever this process would have computed.  Any other name: this process's key.
  JitHash h = jit_hash(src, fma != 0);
  {
    const char* base = strrchr(out_path, '/');
    base = base ? base + 1 : out_path;
    unsigned long long a = 0, b = 0;
    int used = 0;
    if (strlen(base) == 35 && sscanf(base, "%16llx%16llx%n", &a, &b, &used) == 2 && used == 32 && !strcmp(base + 32, ".co")) {
      h.a = a;
      h.b = b;
    }
  }
  if (!jit_disk_write(out_path, src.size(), h, code)) return fail(QIP_ERR_DEVICE, "cannot write %s", out_path);
  return QIP_OK;
} QIP_CATCH_ALL

// the CPUs this process may actually use: affinity mask, capped by the cgroup's CPU quota (the GPU boxes show 256 CPUs and grant 16)
static int usable_cpus() {
  int n = (int)std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::max(1, CPU_COUNT(&set));
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32] = {0};
    long long period = 0;
    if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      const long long quota = atoll(q);
      if (quota > 0) n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
    }
    fclose(f);
  }
  return n;
}
static int jit_procs_effective() {
  if (g_jit_procs >= 1) return (int)std::min<int64_t>(g_jit_procs, 64);
  static const int cpus = usable_cpus();
  return std::max(1, std::min(16, cpus) / (int)std::max<int64_t>(1, g_jit_world));
}
// the helper program: $QIP_HIP_JITC, or `qip_jitc` next to this library
static std::string jit_helper_path() {
  if (const char* e = getenv("QIP_HIP_JITC")) return access(e, X_OK) == 0 ? std::string(e) : std::string();
  Dl_info info;
  if (!dladdr((void*)&qip_hip_jit_compile_file, &info) || !info.dli_fname) return std::string();
  std::string pth = info.dli_fname;
  const size_t slash = pth.find_last_of('/');
  pth = (slash == std::string::npos ? std::string(".") : pth.substr(0, slash)) + "/qip_jitc";
  return access(pth.c_str(), X_OK) == 0 ? pth : std::string();
}
// Compile jobs[todo[*]] in up to `procs` helper processes; results land at out_paths[*] (the disk cache).  Failures are not
// reported from here: whatever is still missing afterwards is compiled in this process, which also produces the message.
static void jit_compile_in_helpers(const std::string& helper, const std::string& dir, int procs,
                                   const std::vector<std::pair<std::string, bool>>& jobs, const std::vector<size_t>& todo,
                                   const std::vector<std::string>& out_paths) {
  std::vector<std::string> src_paths(jobs.size());
  std::vector<size_t> written;
  for (size_t i : todo) {
    char name[64];
    snprintf(name, sizeof name, "/seg.%ld.%u.%zu.hip", (long)getpid(), g_jit_tmp_serial.fetch_add(1), i);
    src_paths[i] = dir + name;
    FILE* f = fopen(src_paths[i].c_str(), "wb");
    if (!f) continue;
    const bool ok = fwrite(jobs[i].first.data(), 1, jobs[i].first.size(), f) == jobs[i].first.size();
    if (fclose(f) == 0 && ok) written.push_back(i);
  }
  const size_t np = std::min<size_t>((size_t)procs, written.size());
  std::vector<pid_t> pids;
  for (size_t k = 0; k < np; ++k) {
    std::vector<std::string> args = {helper};
    for (size_t j = k; j < written.size(); j += np) {  // round robin: heavy and light segments alternate along a plan
      const size_t i = written[j];
      args.push_back(jobs[i].second ? "1" : "0");
      args.push_back(src_paths[i]);
      args.push_back(out_paths[i]);
    }
    std::vector<char*> argv;
    for (std::string& a : args) argv.push_back(&a[0]);
    argv.push_back(nullptr);
    pid_t pid = 0;
    if (posix_spawn(&pid, helper.c_str(), nullptr, nullptr, argv.data(), environ) == 0) pids.push_back(pid);
  }
  for (pid_t pid : pids) {
    int status = 0;
    while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {
    }
  }
  for (size_t i : written) (void)unlink(src_paths[i].c_str());
  g_jit_helper_procs += pids.size();
}

struct JitKernel {
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr;
  uint64_t last_use = 0;
  int device = 0;
};
// key: device ordinal + compile flavour + source text.  Bounded: beyond g_jit_cache_cap entries (global option
// "jit_cache_cap", default 512 — a parameter sweep that changes matrix constants makes new sources without end) the least
// recently used kernels are unloaded, oldest first.  A kernel that is evicted while a captured hipGraph still names it would
// dangle, so programs hold the cache generation they were recorded under and re-record when it moved (qip_hip_program_run).
static std::map<std::string, JitKernel> g_jit_cache;
static uint64_t g_jit_clock = 0, g_jit_generation = 0;
#include <set>
static std::set<uint64_t> g_jit_warm_plans;  // fingerprints of plans whose segments have all been made resident (apply_ops_tiled)
static uint64_t g_jit_warm_generation = 0;   // ... as long as nothing has been evicted since
static int64_t g_jit_cache_cap = 512;
static uint64_t g_jit_compiles = 0, g_jit_evictions = 0, g_jit_loaded = 0;  // (compiles: hiprtc runs of THIS process; loaded: kernels made resident)
static double g_jit_compile_ms = 0;
// Stream captures in progress in this process, on any handle (under g_jit_mutex).  Evicting synchronises the device and
// unloads modules: neither may happen while ANY thread records a graph, not only the evicting handle's own capture.
static int g_jit_captures_in_progress = 0;
static void jit_capture_scope(int delta) {
  std::lock_guard<std::mutex> lock(g_jit_mutex);
  g_jit_captures_in_progress += delta;
}

extern "C" int qip_hip_jit_stats2(qip_hip_jit_counters* out) try {
  if (!out) return fail(QIP_ERR_INVALID, "null output");
  std::lock_guard<std::mutex> lock(g_jit_mutex);
  out->kernels_resident_total = g_jit_loaded;
  out->compiled = g_jit_compiles;
  out->compiled_by_helpers = g_jit_helper_segments;
  out->helper_processes = g_jit_helper_procs;
  out->disk_hits = g_jit_disk_hits;
  out->disk_stores = g_jit_disk_stores;
  out->compile_ms = g_jit_compile_ms;
  out->disk_load_ms = g_jit_load_ms;
  out->procs = jit_procs_effective();
  out->disk_cache = (g_jit_disk && !jit_dir_locked().empty()) ? 1 : 0;
  out->background_segments = g_jit_background_segments;
  out->disk_trimmed = g_jit_disk_trimmed;
  return QIP_OK;
} QIP_CATCH_ALL

// Where code objects are kept.  dir = NULL: back to the default ($QIP_HIP_CACHE_DIR, $XDG_CACHE_HOME/qip_hip, ~/.cache/qip_hip);
// "" : nowhere.  A directory that cannot be created or written is an error and leaves the setting unchanged.
extern "C" int qip_hip_jit_set_cache_dir(const char* dir) try {
  std::lock_guard<std::mutex> lock(g_jit_mutex);
  if (!dir) {
    g_jit_dir_resolved = false;
    (void)jit_dir_locked();
    return QIP_OK;
  }
  if (!*dir) {
    g_jit_dir.clear();
    g_jit_dir_resolved = true;
    return QIP_OK;
  }
  const std::string d = dir;
  std::string why = "cannot be created";
  if (!mkdir_p(d) || !jit_dir_trusted(d, &why))
    return fail(QIP_ERR_INVALID, "cache directory %s is not used: %s (code objects found there are loaded onto the GPU: the directory must belong to this user and be writable by nobody else)", dir, why.c_str());
  g_jit_dir = d;
  g_jit_dir_resolved = true;
  return QIP_OK;
} QIP_CATCH_ALL
extern "C" const char* qip_hip_jit_cache_dir(void) {
  static thread_local std::string out;
  try {
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    out = g_jit_disk ? jit_dir_locked() : std::string();
  } catch (...) {
    return "";
  }
  return out.c_str();
}

extern "C" int qip_hip_jit_cache_info(uint64_t* resident, uint64_t* evicted, uint64_t* cap) try {
  std::lock_guard<std::mutex> lock(g_jit_mutex);
  if (resident) *resident = g_jit_cache.size();
  if (evicted) *evicted = g_jit_evictions;
  if (cap) *cap = (uint64_t)g_jit_cache_cap;
  return QIP_OK;
} QIP_CATCH_ALL

int jit_set_cache_cap(int64_t cap) {
  if (cap < 1) return fail(QIP_ERR_INVALID, "jit_cache_cap must be >= 1");
  std::lock_guard<std::mutex> lock(g_jit_mutex);
  g_jit_cache_cap = cap;
  return QIP_OK;
}
uint64_t jit_cache_generation() {
  std::lock_guard<std::mutex> lock(g_jit_mutex);
  return g_jit_generation;
}

// under g_jit_mutex: unload least recently used kernels until the cache is within its bound
static void jit_evict_locked() {
  while ((int64_t)g_jit_cache.size() > g_jit_cache_cap) {
    auto victim = g_jit_cache.begin();
    for (auto it = g_jit_cache.begin(); it != g_jit_cache.end(); ++it)
      if (it->second.last_use < victim->second.last_use) victim = it;
    int cur = 0;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(victim->second.device);
    (void)hipDeviceSynchronize();  // no launch of it may still be queued
    (void)hipModuleUnload(victim->second.module);
    (void)hipSetDevice(cur);
    g_jit_cache.erase(victim);
    g_jit_evictions += 1;
    g_jit_generation += 1;
  }
}

template <typename T> static std::string fnum(T v) {
  char buf[64];
  if (std::is_same<T, double>::value) snprintf(buf, sizeof buf, "%.17g", (double)v);
  else snprintf(buf, sizeof buf, "%.9gf", (double)v);
  std::string r = buf;
  // a bare integer literal would not be floating point ("1" / "1f")
  if (std::is_same<T, double>::value && r.find_first_of(".eEn") == std::string::npos) r += ".0";
  if (!std::is_same<T, double>::value && r.find_first_of(".eEn") == std::string::npos) r.insert(r.size() - 1, ".0");
  return r;
}

// The segment as HIP source (see the block comment above).  Mirrors k_tile_passes statement by statement.
// (Measured in round 3 and NOT adopted: persistent blocks that prefetch the next tile — into registers, or by LDS-DMA into a
// second tile slot.  The skeleton of a sweep already overlaps its phases through the five resident blocks per CU: with the
// passes' arithmetic scaled from 0 to 6000 f64 operations per lane the time is max(HBM time, issue time) + ~1 ms, and a
// light sweep's time is set by which five high positions the tile holds (5.2 - 6.7 ms), not by the block structure;
// tools/tune_tile.hip, profiles/r03_tile_skeleton.md.)
//
// `params` (option "tile_jit" = 2): the segment's STRUCTURE is compiled, its numbers are kernel data.  Every matrix component
// that is not exactly 0 or +-1 becomes a read of P[k] (a wave-uniform scalar load from the device arena) and its value goes
// to `params`; zeros and units stay literals, and the flags the host derives from them (non-zero mask, real / X shapes) stay
// constants, so the same folding happens and the arithmetic per amplitude is unchanged: bit-identical to "tile_jit" = 1.
// A variational loop that updates its rotation angles produces the same source every time and re-uses the compiled kernel;
// only a value that newly becomes (or stops being) 0 / +-1 changes the structure and compiles again.
template <typename T>
static std::string tile_jit_source(const TileSegmentPlan<T>& plan, const Ins& ins, bool nt, int remap = 0, std::vector<T>* params = nullptr,
                                   bool merge_diag = false, const TileStorePerm* fold = nullptr, bool sliced = false) {
  // `sliced` (r5): the sweep is launched in parts — `ins` has the slice positions opened too, and the kernel's second parameter
  // is the part's bits at those positions (ORed into every block's base) instead of the unused tile count
  const char* tname = std::is_same<T, double>::value ? "double" : "float";
  std::string o;
  auto L = [&](const std::string& line) { o += line; o += "\n"; };
  auto U = [](uint64_t v) { return std::to_string(v) + "ull"; };
  auto comp = [&](T v) -> std::string {
    if (!params || v == (T)0 || v == (T)1 || v == (T)-1) return fnum<T>(v);  // (a -0 prints as -0.0 and stays one)
    params->push_back(v);
    return "P[" + std::to_string(params->size() - 1) + "]";
  };
  auto amp = [&](amp_t<T> a) {
    const std::string re = comp(a.x);  // fixed evaluation order: the parameter numbering is part of the source
    const std::string im = comp(a.y);
    return "{" + re + ", " + im + "}";
  };
  const TilePassDesc& d = plan.pd;
  L("#include \"qip_kernels.h\"");
  L("using namespace qipk;");
  L(std::string("typedef ") + tname + " T;");
  L("typedef amp_t<T> A;");
  // tile number -> amplitude index of the tile's element 0: insert_bits with the positions as literals
  L("__device__ __forceinline__ uint64_t tile_base(uint64_t t) {");
  L("  uint64_t w = t << kTileLow;");
  for (uint32_t j = 0; j < ins.npos; ++j) {
    const std::string p = std::to_string(ins.pos[j]);
    L("  w = ((w >> " + p + ") << " + std::to_string(ins.pos[j] + 1) + ") | (w & ((1ull << " + p + ") - 1ull));");
  }
  if (ins.ormask) L("  w |= " + U(ins.ormask) + ";");
  if (d.p5 != 5u)  // split rows (see tile_block_base): `ins` was opened with 5 standing for p5; the two bits trade places now
    L("  w = (w & ~(1ull << " + std::to_string(d.p5) + ")) | (((w >> " + std::to_string(d.p5) + ") & 1ull) << 5);");
  L("  return w;");
  L("}");
  if (fold && fold->g) {
    // the tiles are stored elsewhere, packed for the multi-GPU exchange (TileStorePerm): the positions as literals
    L("__device__ __forceinline__ uint64_t dst_of(uint64_t x) {");
    L("  uint64_t top = 0;");
    for (uint32_t t = 0; t < fold->g; ++t)
      L("  top |= ((x >> " + std::to_string(fold->sel[t]) + ") & 1ull) << " + std::to_string(fold->Lg + t) + ";");
    for (uint32_t t = 0; t < fold->g; ++t) {
      const std::string p = std::to_string(fold->sel_desc[t]);
      L("  x = ((x >> " + std::to_string(fold->sel_desc[t] + 1) + ") << " + p + ") | (x & ((1ull << " + p + ") - 1ull));");
    }
    L("  return x | top;");
    L("}");
  }
  L(std::string("extern \"C\" __global__ __launch_bounds__(kTileBlock, 5) void qip_segment(A* __restrict__ st, uint64_t ") + (sliced ? "slice_or" : "ntiles") +
    (params ? ", const T* __restrict__ P" : "") + (fold && fold->g ? ", A* __restrict__ out" : "") + ") {");
  L("  extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];");
  L("  A* tile = reinterpret_cast<A*>(tile_raw);");
  L(std::string("  constexpr bool NT = ") + (nt ? "true" : "false") + ";");
  L("  const uint32_t tid = threadIdx.x, lane = tid & 63u;");
  L("  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);");
  L("  uint64_t wave_off = 0;");
  for (int j = 0; j < kTileWaveBits; ++j)
    L("  wave_off |= (uint64_t)((wave >> " + std::to_string(j) + ") & 1u) << " + std::to_string(d.hpos[j]) + ";");
  L("  const uint32_t slot_tid = tile_slot<A>(tid);");
  auto ub = [&](int u) {
    uint64_t off = 0;
    for (int b = 0; b < 3; ++b)
      if ((u >> b) & 1) off |= 1ull << d.hpos[kTileWaveBits + b];
    return U(off);
  };
  L("  A x[8];");
  if (!sliced) L("  (void)ntiles;");
  L("  uint64_t blk = blockIdx.x + (uint64_t)blockIdx.y * gridDim.x;");
  // (tuning aid, global option "tile_remap": the 2^r blocks one XCD receives in a row take ADJACENT tiles; measured slower)
  if (remap == 2) L("  blk = (blk & ~31ull) | ((blk & 7ull) << 2) | ((blk >> 3) & 3ull);");
  if (remap == 3) L("  blk = (blk & ~63ull) | ((blk & 7ull) << 3) | ((blk >> 3) & 7ull);");
  if (remap >= 16) {
    // XCD-aware order (global option "tile_remap" = 4; remap = 16 + J here): consecutive block numbers go round-robin over
    // the 8 XCDs, so block bits 0..2 name the XCD.  They drive three tile-number bits that stand for HIGH free positions
    // (the first three at or above amplitude position 16), the remaining block bits walk the other free positions upwards:
    // each XCD sweeps its own far-apart region with neighbouring 1-KiB rows in consecutive blocks, instead of all eight
    // interleaving on the same rows.  Tuning aid, off by default: on the bare sweep skeleton (tools/tune_tile "order") it is
    // worth 2 - 13 % on tiles whose five positions are scattered, on real segments it is not (configs[1] 69.3 -> 69.7 ms,
    // QFT 36.3 -> 37.9 ms in the HBM-bound tile mode: profiles/r03_tile_skeleton.md, section 4).
    const std::string J = std::to_string(remap - 16);
    L("  blk = ((blk >> 3) & ((1ull << " + J + ") - 1ull)) | ((blk & 7ull) << " + J + ") | (((blk >> 3) >> " + J + ") << (" + J + " + 3));");
  }
  L(std::string("  const uint64_t base = tile_base(blk)") + (sliced ? " | slice_or" : "") + ", wbase = base | wave_off;");
  L("  const uint32_t lane_off = tile_lane_off(lane, " + std::to_string(d.p5) + "u);");
  for (int u = 0; u < 8; ++u) L("  x[" + std::to_string(u) + "] = ldg<NT>(st + (wbase | " + ub(u) + ") + lane_off);");
  L("  const uint32_t tidv = tid;");
  for (int u = 0; u < 8; ++u)
    L("    tile[slot_tid ^ tile_slot<A>(" + std::to_string(u) + "u << kTileLaneBits)] = x[" + std::to_string(u) + "];");
  L("  __syncthreads();");
  for (uint32_t pi = 0; pi < d.npasses; ++pi) {
    const TilePass& ps = d.pass[pi];
    L("  {  // pass " + std::to_string(pi));
    // the pass's LDS addresses depend on the lane id alone: left alone hipcc computes those of EVERY pass up front and keeps
    // them live (spills in segments of many passes); an opaque copy of the id per pass pins them behind the barrier before it
    L("    uint32_t tidp = tidv;");
    L("    asm volatile(\"\" : \"+v\"(tidp));");
    L("    uint32_t tb = 0;");
    for (int k = 0; k < kTileLaneBits; ++k)
      L("    tb |= ((tidp >> " + std::to_string(k) + ") & 1u) << " + std::to_string((unsigned)((ps.lanepos >> (4 * k)) & 15ull)) + ";");
    L("    const uint32_t slot_tb = tile_slot<A>(tb);");
    std::string cs = "    const uint32_t c[8] = {";
    for (int i = 0; i < 8; ++i) {
      const uint32_t c = ((uint32_t)(i & 1) << ps.pb[0]) | ((uint32_t)((i >> 1) & 1) << ps.pb[1]) | ((uint32_t)((i >> 2) & 1) << ps.pb[2]);
      cs += std::to_string(c) + "u" + (i < 7 ? ", " : "};");
    }
    L(cs);
    L("    A e[8];");
    for (int i = 0; i < 8; ++i) L("    e[" + std::to_string(i) + "] = tile[slot_tb ^ tile_slot<A>(c[" + std::to_string(i) + "])];");
    for (uint32_t gi = ps.first; gi < ps.first + ps.count; ++gi) {
      const TileGate<T>& g = plan.gates[gi];
      // `merge_diag` (option "tile_merge", tile = 2 only: 1e-12 bar): a RUN of consecutive diagonal gates — they all commute —
      // is applied as products.  Every gate contributes a factor (per lane: its lane-bit and outside-the-tile conditions
      // select between its entry and 1) to the SET of the lane's eight elements its register-bit conditions pick; factors of
      // one set are multiplied together first, then each element takes the product of its sets: QFT's 29 controlled phases
      // after an H cost ~29 + 4 complex products per lane instead of 4 x 29.
      if (merge_diag && g.kind == 1) {
        uint32_t ge = gi;
        while (ge < ps.first + ps.count && plan.gates[ge].kind == 1) ++ge;
        if (ge - gi >= 3) {
          std::vector<uint32_t> sets;  // element masks, in order of first appearance: F<k> is the running product of set k
          L("    {  // gates " + std::to_string(gi) + " .. " + std::to_string(ge - 1) + ": one run of diagonal gates");
          // (each factor joins its product at once: short live ranges.  `ucond`: a block-uniform condition — a control outside the
          // tile — stays a BRANCH around the product, as in the gate-by-gate form: no work at all where the control reads 0)
          auto add = [&](uint32_t mask, const std::string& expr, const std::string& ucond) {
            if (!mask) return;
            const std::string open = ucond.empty() ? "" : "if (" + ucond + ") { QIP_KEEP_BRANCH(); ", close = ucond.empty() ? "" : " }";
            for (size_t k = 0; k < sets.size(); ++k)
              if (sets[k] == mask) {
                L("      " + open + "F" + std::to_string(k) + " = cmul(F" + std::to_string(k) + ", " + expr + ");" + close);
                return;
              }
            if (ucond.empty()) L("      A F" + std::to_string(sets.size()) + " = " + expr + ";");
            else L("      A F" + std::to_string(sets.size()) + " = {(T)1, (T)0}; " + open + "F" + std::to_string(sets.size()) + " = " + expr + ";" + close);
            sets.push_back(mask);
          };
          for (uint32_t gj = gi; gj < ge; ++gj) {
            const TileGate<T>& d = plan.gates[gj];
            uint32_t ok = 0;  // elements whose register-bit controls are all 1
            for (int i = 0; i < 8; ++i) {
              const uint32_t ci = ((uint32_t)(i & 1) << ps.pb[0]) | ((uint32_t)((i >> 1) & 1) << ps.pb[1]) | ((uint32_t)((i >> 2) & 1) << ps.pb[2]);
              if ((ci & d.cm_reg) == d.cm_reg) ok |= 1u << i;
            }
            const std::string m0 = amp(d.m[0]), m1 = amp(d.m[1]);
            const bool u0 = d.m[0].x == (T)1 && d.m[0].y == (T)0, u1 = d.m[1].x == (T)1 && d.m[1].y == (T)0;
            // the conditions every element of the lane shares: controls outside the tile (block-uniform: a branch) and on lane
            // bits (a select between the entry and 1)
            const std::string ucond = d.omask ? "(base & " + U(d.omask) + ") == " + U(d.omask) : "";
            const std::string lcond = d.cm_lane ? "((tb & " + std::to_string(d.cm_lane) + "u) == " + std::to_string(d.cm_lane) + "u)" : "";
            auto guarded = [&](const std::string& f) { return lcond.empty() ? f : "tile_sel(" + lcond + ", " + f + ", A{(T)1, (T)0})"; };
            if (d.op >= TOP_DIAG_REG0 && d.op <= TOP_DIAG_REG2) {
              const int J = (int)(d.op - TOP_DIAG_REG0);
              uint32_t half1 = 0;
              for (int i = 0; i < 8; ++i)
                if ((i >> J) & 1) half1 |= 1u << i;
              if (!u0) add(ok & ~half1, guarded("A" + m0), ucond);
              if (!u1) add(ok & half1, guarded("A" + m1), ucond);
            } else {  // the target is a lane bit or lies outside the tile: one factor for every element the controls pick
              const std::string one = d.b0 == kTileOutside ? "(((base >> " + std::to_string(d.tpos_out) + ") & 1ull) != 0)"
                                                           : "(((tb >> " + std::to_string(d.b0) + ") & 1u) != 0)";
              add(ok, guarded("tile_sel(" + one + ", A" + m1 + ", A" + m0 + ")"), ucond);
            }
          }
          for (int i = 0; i < 8; ++i)
            for (size_t si = 0; si < sets.size(); ++si)
              if ((sets[si] >> i) & 1u) L("      e[" + std::to_string(i) + "] = cmul(F" + std::to_string(si) + ", e[" + std::to_string(i) + "]);");
          L("    }");
          gi = ge - 1;
          continue;
        }
      }
      L("    {  // gate " + std::to_string(gi));
      {
        const std::string m0 = amp(g.m[0]), m1 = amp(g.m[1]), m2 = amp(g.m[2]), m3 = amp(g.m[3]);
        L(std::string("      ") + (params ? "const" : "constexpr") + " TileGate<T> g = {" + std::to_string(g.kind) + "u, " + std::to_string(g.b0) + "u, " +
          std::to_string(g.b1) + "u, " + std::to_string(g.cmask) + "u, " + std::to_string(g.nz) + "u, " + std::to_string(g.tpos_out) + "u, " + U(g.omask) +
          ", " + std::to_string(g.op) + "u, " + std::to_string(g.cm_reg) + "u, " + std::to_string(g.cm_lane) + "u, 0u, {" + m0 + ", " + m1 + ", " + m2 +
          ", " + m3 + "}};");
      }
      const std::string lane_args = "g.cm_lane != 0u, (tb & g.cm_lane) == g.cm_lane";
      std::string call;
      switch (g.op) {
        // Diagonal gates: which entries are the unit is known HERE (a unit is always written as a literal, also in the
        // parametrised form), so the helpers' run-time "unit entries leave the amplitude untouched" tests are resolved by
        // the generator: an entry that is the unit emits nothing.  Per amplitude the products are those of pass_diag / the
        // interpreter's switch, in the same order.
        case TOP_DIAG_UNIFORM: {
          const bool u0 = g.m[0].x == (T)1 && g.m[0].y == (T)0, u1 = g.m[1].x == (T)1 && g.m[1].y == (T)0;
          const std::string s0 = u0 ? "" : "pass_scale<T, 0, -1>(g.m[0], e, c, g.cm_reg);", s1 = u1 ? "" : "pass_scale<T, 0, -1>(g.m[1], e, c, g.cm_reg);";
          call = "if ((base >> g.tpos_out) & 1ull) { " + s1 + " } else { " + s0 + " }";
          break;
        }
        case TOP_DIAG_LANE:
        case TOP_DIAG_LANE_CTL:
          call = std::string("const bool one = ") + (g.b0 == kTileOutside ? "((base >> g.tpos_out) & 1ull) != 0" : "((tb >> g.b0) & 1u) != 0") +
                 "; A f = tile_sel(one, g.m[1], g.m[0]); " +
                 (g.op == TOP_DIAG_LANE_CTL ? "{ const bool lane_ok = (tb & g.cm_lane) == g.cm_lane; f.x = lane_ok ? f.x : (T)1; f.y = lane_ok ? f.y : (T)0; } " : "") +
                 "pass_scale<T, 0, -1>(f, e, c, g.cm_reg);";
          break;
        case TOP_DIAG_REG0: case TOP_DIAG_REG1: case TOP_DIAG_REG2: {
          const std::string J = std::to_string(g.op - TOP_DIAG_REG0);
          call.clear();
          for (int half = 0; half < 2; ++half) {
            if (g.m[half].x == (T)1 && g.m[half].y == (T)0) continue;
            const std::string hs = std::to_string(half);
            call += "{ A f = g.m[" + hs + "]; ";
            if (g.cm_lane) call += "{ const bool lane_ok = (tb & g.cm_lane) == g.cm_lane; f.x = lane_ok ? f.x : (T)1; f.y = lane_ok ? f.y : (T)0; } ";
            call += "pass_scale<T, " + J + ", " + hs + ">(f, e, c, g.cm_reg); } ";
          }
          break;
        }
        case TOP_DENSE0: case TOP_DENSE1: case TOP_DENSE2:
          call = "pass_dense<T, " + std::to_string(g.op - TOP_DENSE0) + ">(g, e, c, g.cm_reg);";
          break;
        case TOP_DENSE_LANE0: case TOP_DENSE_LANE1: case TOP_DENSE_LANE2:
          call = "pass_dense_lane<T, " + std::to_string(g.op - TOP_DENSE_LANE0) + ">(g, e, c, g.cm_reg, (tb & g.cm_lane) == g.cm_lane);";
          break;
        case TOP_DENSE2Q_01: case TOP_DENSE2Q_02: case TOP_DENSE2Q_10: case TOP_DENSE2Q_12: case TOP_DENSE2Q_20: case TOP_DENSE2Q_21: {
          static const int ja[6] = {0, 0, 1, 1, 2, 2}, jb[6] = {1, 2, 0, 2, 0, 1};
          std::string m = "const A M[16] = {";
          for (int e = 0; e < 16; ++e) m += amp(plan.mats[16 * g.nz + e]) + (e < 15 ? ", " : "}; ");
          call = m + "pass_dense2<T, " + std::to_string(ja[g.op - TOP_DENSE2Q_01]) + ", " + std::to_string(jb[g.op - TOP_DENSE2Q_01]) + ">(M, e, c, g.cm_reg, " + lane_args + ");";
          break;
        }
        case TOP_DENSE3Q_012: case TOP_DENSE3Q_021: case TOP_DENSE3Q_102: case TOP_DENSE3Q_120: case TOP_DENSE3Q_201: case TOP_DENSE3Q_210: {
          static const int ja[6] = {0, 0, 1, 1, 2, 2}, jb[6] = {1, 2, 0, 2, 0, 1};
          const int a = ja[g.op - TOP_DENSE3Q_012], b = jb[g.op - TOP_DENSE3Q_012];
          const std::string tail = "pass_dense3<T, " + std::to_string(a) + ", " + std::to_string(b) + ", " + std::to_string(3 - a - b) + ">(M, e, " + lane_args + ");";
          if (params) {
            // r4: the 8 x 8 matrix as ONE contiguous block of the parameter array, read row by row through a pointer like the
            // interpreter does (wave-uniform scalar loads as they are needed).  Spelt out as 64 named components, all 128
            // scalars were live at once: the dense-k3 Grover segments ran SLOWER compiled than interpreted (108.9 vs 101.4 ms).
            if (params->size() & 1) params->push_back((T)0);  // (16-byte alignment of the block for Complex<f64>)
            const size_t off = params->size();
            for (int e = 0; e < 64; ++e) {
              params->push_back(plan.mats[16 * g.nz + e].x);
              params->push_back(plan.mats[16 * g.nz + e].y);
            }
            call = "const A* __restrict__ M = reinterpret_cast<const A*>(P + " + std::to_string(off) + "); " + tail;
          } else {
            std::string m = "const A M[64] = {";
            for (int e = 0; e < 64; ++e) m += amp(plan.mats[16 * g.nz + e]) + (e < 63 ? ", " : "}; ");
            call = m + tail;
          }
          break;
        }
        case TOP_SWAP_01: call = "pass_swap<T, 0, 1>(e, c, g.cm_reg, " + lane_args + ");"; break;
        case TOP_SWAP_02: call = "pass_swap<T, 0, 2>(e, c, g.cm_reg, " + lane_args + ");"; break;
        case TOP_SWAP_12: call = "pass_swap<T, 1, 2>(e, c, g.cm_reg, " + lane_args + ");"; break;
        default: break;
      }
      if (g.omask) L("      if ((base & g.omask) == g.omask) { " + call + " }");  // an outside control is 0 for this whole tile
      else L("      { " + call + " }");
      // one gate at a time: left alone hipcc interleaves neighbouring gates up to the register budget of the launch bound and
      // then spills (16 - 104 bytes of scratch per lane in the configs[1] segments, 104 vs 90 ms for the circuit)
      L("      __builtin_amdgcn_sched_barrier(0);");
      L("    }");
    }
    for (int i = 0; i < 8; ++i) L("    tile[slot_tb ^ tile_slot<A>(c[" + std::to_string(i) + "])] = e[" + std::to_string(i) + "];");
    L("    __syncthreads();");
    L("  }");
  }
  for (int u = 0; u < 8; ++u)
    if (fold && fold->g)
      L("  stg<NT>(out + dst_of(wbase | " + ub(u) + ") + dst_of(lane_off), tile[slot_tid ^ tile_slot<A>(" + std::to_string(u) + "u << kTileLaneBits)]);");
    else
      L("  stg<NT>(st + (wbase | " + ub(u) + ") + lane_off, tile[slot_tid ^ tile_slot<A>(" + std::to_string(u) + "u << kTileLaneBits)]);");
  L("}");
  return o;
}

// ---- wide tiles (r4, option "tile_wide"): the segment's source for the register-resident 13-bit tile ------------------------------
// One array of 32 amplitudes per arrangement (straight-line code with constant indices: the arrays are names, not memory — a
// transposition costs its LDS traffic and two barriers per quarter, no register moves); the gates go through the same helpers
// as the 11-bit sweeps (pass_dense, pass_scale, pass_swap, pass_dense2, pass_dense3w over 32 elements: the products and sums of
// the gate-by-gate kernels in the same order — a circuit-order segment stays IEEE-equal to them).
// `pin` (global option "tile_wide_pin", debug mode bit 256; r4, profiles/r04_wide_tiles.md): after every gate that
// is applied under a block-uniform branch (a control / selector outside the tile) the 32 amplitudes pass through an empty asm with
// "+v" constraints.  Semantically nothing; it stops the register allocator from keeping both versions of the tile alive across the join:
// Clifford+T's first segment 175 VGPR spills -> 0 (9238 -> 8646 instructions), QFT's 32 -> 0; configs[1] (no spills) + 4 % instructions.
// On the GPU (tools/exp_wide_pin.py, n = 30, tile = 1, medians of 5): a 72-gate Clifford+T prefix 25.92 -> 23.90 ms (-7.8 %), a 60-gate
// configs[1] prefix 17.42 -> 17.49 ms (noise); results bit-identical.  Default on.
int64_t g_tile_wide_pin = 1;
// global option "tile_wide_dense3_inline" (debug mode bit 512; r4, OFF: compiled and register-checked without a GPU, never run on one):
// dense 3-qubit gates of a wide segment written out group by group instead of through pass_dense3w — the dense-k3 Grover variant's first
// wide segment: 528 B of stack per lane -> 0 (tools/jit_segment_resources.py, mode | 512).  The next round's first measurement.
int64_t g_tile_wide_dense3_inline = 1;  // r5: run on MI355X — bit-identical to pass_dense3w, dense-k3 Grover on wide tiles 109.9 -> 77.5 ms (narrow: 88.2): on
template <typename T>
static std::string wide_jit_source(const WidePlan<T>& plan, const Ins& ins, bool nt, std::vector<T>* params, bool merge_diag = false, bool pin = false,
                                   bool dense3_inline = false, bool sliced = false, const TileStorePerm* fold = nullptr) {
  const char* tname = std::is_same<T, double>::value ? "double" : "float";
  constexpr uint32_t SW = sizeof(amp_t<T>) == 16 ? 4u : 5u;  // tile_slot's fold width
  auto slot = [&](uint32_t t) { return t ^ ((t >> SW) & ((1u << SW) - 1u)); };
  std::string o;
  auto L = [&](const std::string& line) { o += line; o += "\n"; };
  auto U = [](uint64_t v) { return std::to_string(v) + "ull"; };
  auto N = [](uint64_t v) { return std::to_string(v); };
  auto comp = [&](T v) -> std::string {
    if (!params || v == (T)0 || v == (T)1 || v == (T)-1) return fnum<T>(v);
    params->push_back(v);
    return "P[" + std::to_string(params->size() - 1) + "]";
  };
  auto amp = [&](amp_t<T> a) {
    const std::string re = comp(a.x);
    const std::string im = comp(a.y);
    return "{" + re + ", " + im + "}";
  };
  L("#include \"qip_kernels.h\"");
  L("using namespace qipk;");
  L(std::string("typedef ") + tname + " T;");
  L("typedef amp_t<T> A;");
  L("__device__ __forceinline__ uint64_t tile_base(uint64_t t) {");
  L("  uint64_t w = t << kTileLow;");
  for (uint32_t j = 0; j < ins.npos; ++j) {
    const std::string p = N(ins.pos[j]);
    L("  w = ((w >> " + p + ") << " + N(ins.pos[j] + 1) + ") | (w & ((1ull << " + p + ") - 1ull));");
  }
  if (plan.p5 != 5u) L("  w = (w & ~(1ull << " + N(plan.p5) + ")) | (((w >> " + N(plan.p5) + ") & 1ull) << 5);");
  L("  return w;");
  L("}");
  if (fold && fold->g) {
    // r5: the tiles are stored elsewhere, packed for the multi-GPU exchange (TileStorePerm), exactly as the 11-bit sweeps do it: the
    // map moves index bits, so it distributes over wave base | access bits | lane offset
    L("__device__ __forceinline__ uint64_t dst_of(uint64_t x) {");
    L("  uint64_t top = 0;");
    for (uint32_t t = 0; t < fold->g; ++t) L("  top |= ((x >> " + N(fold->sel[t]) + ") & 1ull) << " + N(fold->Lg + t) + ";");
    for (uint32_t t = 0; t < fold->g; ++t) {
      const std::string p = N(fold->sel_desc[t]);
      L("  x = ((x >> " + N(fold->sel_desc[t] + 1) + ") << " + p + ") | (x & ((1ull << " + p + ") - 1ull));");
    }
    L("  return x | top;");
    L("}");
  }
  L(std::string("extern \"C\" __global__ __launch_bounds__(256, 2) void qip_segment(A* __restrict__ st, uint64_t ") + (sliced ? "slice_or" : "ntiles") +
    (params ? ", const T* __restrict__ P" : "") + (fold && fold->g ? ", A* __restrict__ out" : "") + ") {");
  L("  extern __shared__ __attribute__((aligned(16))) unsigned char buf_raw[];");
  L("  A* buf = reinterpret_cast<A*>(buf_raw);");
  L(std::string("  constexpr bool NT = ") + (nt ? "true" : "false") + ";");
  L("  const uint32_t tid = threadIdx.x, lane = tid & 63u;");
  L("  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);");
  if (!sliced) L("  (void)ntiles;");
  L(std::string("  const uint64_t base = tile_base(blockIdx.x + (uint64_t)blockIdx.y * gridDim.x)") + (sliced ? " | slice_or;" : ";"));
  L("  const uint64_t wbase = base | ((uint64_t)(wave & 1u) << " + N(plan.high[0]) + ") | ((uint64_t)(wave >> 1) << " + N(plan.high[1]) + ");");
  L("  const uint32_t lane_off = tile_lane_off(lane, " + N(plan.p5) + "u);");
  auto ub = [&](int u) {
    uint64_t off = 0;
    for (int b = 0; b < kWideRegBits; ++b)
      if ((u >> b) & 1) off |= 1ull << plan.high[2 + b];
    return U(off);
  };
  L("  A e0[32];");
  for (int u = 0; u < 32; ++u) L("  e0[" + N(u) + "] = ldg<NT>(st + (wbase | " + ub(u) + ") + lane_off);");
  L("  const uint32_t tidv = tid;");
  for (size_t pi = 0; pi < plan.passes.size(); ++pi) {
    const WidePass& ps = plan.passes[pi];
    const std::string e = "e" + N(pi);
    auto jof = [&](uint32_t bit) {
      for (int j = 0; j < kWideRegBits; ++j)
        if (ps.R[j] == bit) return j;
      return -1;
    };
    if (ps.transposed) {
      const WidePass& pa = plan.passes[pi - 1];
      const std::string ea = "e" + N(pi - 1);
      L("  A " + e + "[32];");
      L("  {  // transposition into arrangement " + N(pi));
      L("    uint32_t tidp = tidv;");
      L("    asm volatile(\"\" : \"+v\"(tidp));");
      L("    uint32_t wa = 0, rb = 0;");
      for (int k = 0; k < 8; ++k) {
        L("    wa |= ((tidp >> " + N(k) + ") & 1u) << " + N(ps.bufpos[pa.L[k]]) + ";");
        L("    rb |= ((tidp >> " + N(k) + ") & 1u) << " + N(ps.bufpos[ps.L[k]]) + ";");
      }
      L("    const uint32_t swa = tile_slot<A>(wa), srb = tile_slot<A>(rb);");
      auto split = [&](const WidePass& w, int (&jq)[2], int (&jo)[3]) {
        int no = 0;
        for (int j = 0; j < kWideRegBits; ++j) {
          if (w.R[j] == ps.q[0]) jq[0] = j;
          else if (w.R[j] == ps.q[1]) jq[1] = j;
          else jo[no++] = j;
        }
      };
      int jqa[2] = {0, 0}, joa[3] = {0, 0, 0}, jqb[2] = {0, 0}, job[3] = {0, 0, 0};
      split(pa, jqa, joa);
      split(ps, jqb, job);
      // two buffer halves: quarter v + 1 is written while quarter v is read (write 0 | barrier | read 0, write 1 | barrier |
      // read 1, write 2 | ... ): four barriers per transposition instead of seven
      auto emit_write = [&](int qv) {
        for (int m = 0; m < 8; ++m) {
          uint32_t r = ((uint32_t)(qv & 1) << jqa[0]) | ((uint32_t)(qv >> 1) << jqa[1]), ci = 0;
          for (int i = 0; i < 3; ++i)
            if ((m >> i) & 1) {
              r |= 1u << joa[i];
              ci |= 1u << ps.bufpos[pa.R[joa[i]]];
            }
          L("    buf[" + N((qv & 1) * 2048) + "u + (swa ^ " + N(slot(ci)) + "u)] = " + ea + "[" + N(r) + "];");
        }
      };
      auto emit_read = [&](int qv) {
        for (int m = 0; m < 8; ++m) {
          uint32_t r = ((uint32_t)(qv & 1) << jqb[0]) | ((uint32_t)(qv >> 1) << jqb[1]), ci = 0;
          for (int i = 0; i < 3; ++i)
            if ((m >> i) & 1) {
              r |= 1u << job[i];
              ci |= 1u << ps.bufpos[ps.R[job[i]]];
            }
          L("    " + e + "[" + N(r) + "] = buf[" + N((qv & 1) * 2048) + "u + (srb ^ " + N(slot(ci)) + "u)];");
        }
      };
      emit_write(0);
      for (int qv = 0; qv < 4; ++qv) {
        L("    __syncthreads();");
        emit_read(qv);
        if (qv < 3) emit_write(qv + 1);
      }
      L("  }");
      // (no barrier before the next transposition: its first write goes to half 0, whose last reads — quarter 2 — lie before
      // the fourth barrier above; half 1 is next written after the next transposition's first barrier, behind everybody's
      // reads of quarter 3)
    } else if (pi > 0) {
      return std::string();  // (cannot happen: every pass after the first is a transposition)
    }
    if (ps.count == 0) continue;
    L("  {  // pass " + N(pi) + ": register bits = tile bits " + N(ps.R[0]) + "," + N(ps.R[1]) + "," + N(ps.R[2]) + "," + N(ps.R[3]) + "," + N(ps.R[4]));
    L("    uint32_t tidq = tidv;");
    L("    asm volatile(\"\" : \"+v\"(tidq));");
    L("    uint32_t tb = 0;");
    for (int k = 0; k < 8; ++k) L("    tb |= ((tidq >> " + N(k) + ") & 1u) << " + N(ps.L[k]) + ";");
    uint32_t rmask = 0;
    for (int j = 0; j < kWideRegBits; ++j) rmask |= 1u << ps.R[j];
    std::string cs = "    const uint32_t c[32] = {";
    for (int i = 0; i < 32; ++i) {
      uint32_t c = 0;
      for (int j = 0; j < kWideRegBits; ++j)
        if ((i >> j) & 1) c |= 1u << ps.R[j];
      cs += N(c) + "u" + (i < 31 ? ", " : "};");
    }
    L(cs);
    L("    A (&e)[32] = " + e + ";");
    for (uint32_t gi = ps.first; gi < ps.first + ps.count; ++gi) {
      TileGate<T> g = plan.gates[gi];
      g.cm_reg = g.cmask & rmask;
      g.cm_lane = g.cmask & ~rmask;
      g.op = 0;
      // `merge_diag` (option "tile_merge", tile = 2 only: 1e-12 bar): a run of consecutive diagonal gates as products — every
      // gate contributes a factor to the SET of the lane's 32 elements its register-bit conditions pick, the factors of one set
      // are multiplied together as they come, each element then takes the product of its sets (the 11-bit generator's scheme
      // over 32 elements: QFT's runs of controlled phases)
      if (merge_diag && g.kind == 1) {
        uint32_t ge = gi;
        while (ge < ps.first + ps.count && plan.gates[ge].kind == 1) ++ge;
        if (ge - gi >= 3) {
          std::vector<uint32_t> sets;
          L("    {  // gates " + N(gi) + " .. " + N(ge - 1) + ": one run of diagonal gates");
          auto add = [&](uint32_t mask, const std::string& expr, const std::string& ucond) {
            if (!mask) return;
            const std::string open = ucond.empty() ? "" : "if (" + ucond + ") { QIP_KEEP_BRANCH(); ", close = ucond.empty() ? "" : " }";
            for (size_t k = 0; k < sets.size(); ++k)
              if (sets[k] == mask) {
                L("      " + open + "F" + N(k) + " = cmul(F" + N(k) + ", " + expr + ");" + close);
                return;
              }
            if (ucond.empty()) L("      A F" + N(sets.size()) + " = " + expr + ";");
            else L("      A F" + N(sets.size()) + " = {(T)1, (T)0}; " + open + "F" + N(sets.size()) + " = " + expr + ";" + close);
            sets.push_back(mask);
          };
          for (uint32_t gj = gi; gj < ge; ++gj) {
            const TileGate<T>& d = plan.gates[gj];
            const uint32_t d_reg = d.cmask & rmask, d_lane = d.cmask & ~rmask;
            uint32_t ok = 0;  // elements whose register-bit controls are all 1
            for (int i = 0; i < 32; ++i) {
              uint32_t ci = 0;
              for (int j = 0; j < kWideRegBits; ++j)
                if ((i >> j) & 1) ci |= 1u << ps.R[j];
              if ((ci & d_reg) == d_reg) ok |= 1u << i;
            }
            const std::string m0 = amp(d.m[0]), m1 = amp(d.m[1]);
            const bool u0 = d.m[0].x == (T)1 && d.m[0].y == (T)0, u1 = d.m[1].x == (T)1 && d.m[1].y == (T)0;
            const std::string ucond = d.omask ? "(base & " + U(d.omask) + ") == " + U(d.omask) : "";
            const std::string lcond = d_lane ? "((tb & " + N(d_lane) + "u) == " + N(d_lane) + "u)" : "";
            auto guarded = [&](const std::string& f) { return lcond.empty() ? f : "tile_sel(" + lcond + ", " + f + ", A{(T)1, (T)0})"; };
            const int J = d.b0 == kTileOutside ? -1 : jof(d.b0);
            if (J >= 0) {
              uint32_t half1 = 0;
              for (int i = 0; i < 32; ++i)
                if ((i >> J) & 1) half1 |= 1u << i;
              if (!u0) add(ok & ~half1, guarded("A" + m0), ucond);
              if (!u1) add(ok & half1, guarded("A" + m1), ucond);
            } else {
              const std::string one = d.b0 == kTileOutside ? "(((base >> " + N(d.tpos_out) + ") & 1ull) != 0)" : "(((tb >> " + N(d.b0) + ") & 1u) != 0)";
              add(ok, guarded("tile_sel(" + one + ", A" + m1 + ", A" + m0 + ")"), ucond);
            }
          }
          for (int i = 0; i < 32; ++i)
            for (size_t si = 0; si < sets.size(); ++si)
              if ((sets[si] >> i) & 1u) L("      e[" + N(i) + "] = cmul(F" + N(si) + ", e[" + N(i) + "]);");
          L("    }");
          gi = ge - 1;
          continue;
        }
      }
      L("    {  // gate " + N(gi));
      {
        const std::string m0 = amp(g.m[0]), m1 = amp(g.m[1]), m2 = amp(g.m[2]), m3 = amp(g.m[3]);
        L(std::string("      ") + (params ? "const" : "constexpr") + " TileGate<T> g = {" + N(g.kind) + "u, " + N(g.b0) + "u, " + N(g.b1) + "u, " + N(g.cmask) + "u, " +
          N(g.nz) + "u, " + N(g.tpos_out) + "u, " + U(g.omask) + ", 0u, " + N(g.cm_reg) + "u, " + N(g.cm_lane) + "u, 0u, {" + m0 + ", " + m1 + ", " + m2 + ", " + m3 + "}};");
      }
      const std::string lane_args = "g.cm_lane != 0u, (tb & g.cm_lane) == g.cm_lane";
      auto matrix = [&](int cnt) {  // the gate's 4x4 / 8x8 matrix: a block of the parameter array, or literals
        if (params) {
          if (params->size() & 1) params->push_back((T)0);
          const size_t off = params->size();
          for (int k = 0; k < cnt; ++k) {
            params->push_back(plan.mats[16 * g.nz + k].x);
            params->push_back(plan.mats[16 * g.nz + k].y);
          }
          return "const A* __restrict__ M = reinterpret_cast<const A*>(P + " + N(off) + "); ";
        }
        std::string m = "const A M[" + N(cnt) + "] = {";
        for (int k = 0; k < cnt; ++k) m += amp(plan.mats[16 * g.nz + k]) + (k + 1 < cnt ? ", " : "}; ");
        return m;
      };
      std::string call;
      if (g.kind == 1) {
        const bool u0 = g.m[0].x == (T)1 && g.m[0].y == (T)0, u1 = g.m[1].x == (T)1 && g.m[1].y == (T)0;
        const int J = g.b0 == kTileOutside ? -1 : jof(g.b0);
        const std::string guard = g.cm_lane ? "{ const bool lane_ok = (tb & g.cm_lane) == g.cm_lane; f.x = lane_ok ? f.x : (T)1; f.y = lane_ok ? f.y : (T)0; } " : "";
        if (J >= 0) {
          for (int half = 0; half < 2; ++half) {
            if (half == 0 ? u0 : u1) continue;
            call += "{ A f = g.m[" + N(half) + "]; " + guard + "pass_scale<T, " + N(J) + ", " + N(half) + ">(f, e, c, g.cm_reg); } ";
          }
        } else if (g.b0 == kTileOutside && !g.cm_lane) {
          const std::string s0 = u0 ? "" : "pass_scale<T, 0, -1>(g.m[0], e, c, g.cm_reg);", s1 = u1 ? "" : "pass_scale<T, 0, -1>(g.m[1], e, c, g.cm_reg);";
          call = "if ((base >> g.tpos_out) & 1ull) { " + s1 + " } else { " + s0 + " }";
        } else {
          call = std::string("const bool one = ") + (g.b0 == kTileOutside ? "((base >> g.tpos_out) & 1ull) != 0" : "((tb >> g.b0) & 1u) != 0") +
                 "; A f = tile_sel(one, g.m[1], g.m[0]); " + guard + "pass_scale<T, 0, -1>(f, e, c, g.cm_reg);";
        }
      } else if (g.kind == 0) {
        const int J = jof(g.b0);
        call = g.cm_lane ? "pass_dense_lane<T, " + N(J) + ">(g, e, c, g.cm_reg, (tb & g.cm_lane) == g.cm_lane);"
                         : "pass_dense<T, " + N(J) + ">(g, e, c, g.cm_reg);";
      } else if (g.kind == 2) {
        call = "pass_swap<T, " + N(jof(g.b0)) + ", " + N(jof(g.b1)) + ">(e, c, g.cm_reg, " + lane_args + ");";
      } else if (g.kind == 3) {
        call = matrix(16) + "pass_dense2<T, " + N(jof(g.b0)) + ", " + N(jof(g.b1)) + ">(M, e, c, g.cm_reg, " + lane_args + ");";
      } else if (g.kind == 4 && dense3_inline) {
        // The same fold as pass_dense3w, written out for the (at most four) groups this gate really touches, every index a literal.
        // pass_dense3w's loop over the 32 elements is not unrolled by the compiler at NE = 32: `e` is indexed at run time and the whole
        // tile of the lane lives in a 528-byte stack object (profiles/r04_jit_segment_resources.txt: dense-k3 Grover on wide tiles).
        const int JA = jof(g.b0), JB = jof(g.b1), JC = jof(g.tpos_out);
        call = matrix(64);
        for (int base = 0; base < 32; ++base) {
          if (((base >> JA) & 1) || ((base >> JB) & 1) || ((base >> JC) & 1)) continue;
          uint32_t cb = 0;
          for (int j = 0; j < kWideRegBits; ++j)
            if ((base >> j) & 1) cb |= 1u << ps.R[j];
          if ((cb & g.cm_reg) != g.cm_reg) continue;
          auto idx = [&](int r) { return base | (((r >> 2) & 1) << JA) | (((r >> 1) & 1) << JB) | ((r & 1) << JC); };
          call += "{ const A x[8] = {";
          for (int r = 0; r < 8; ++r) call += "e[" + N(idx(r)) + "]" + (r < 7 ? ", " : "}; ");
          for (int r = 0; r < 8; ++r) {
            call += "{ A acc = cmul(M[" + N(r * 8) + "], x[0]); ";
            for (int c2 = 1; c2 < 8; ++c2) call += "acc = cadd(acc, cmul(M[" + N(r * 8 + c2) + "], x[" + N(c2) + "])); ";
            call += "e[" + N(idx(r)) + "] = (g.cm_lane != 0u) ? tile_sel((tb & g.cm_lane) == g.cm_lane, acc, x[" + N(r) + "]) : acc; } __builtin_amdgcn_sched_barrier(0); ";
          }
          call += "} ";
        }
      } else if (g.kind == 4) {
        call = matrix(64) + "pass_dense3w<T, " + N(jof(g.b0)) + ", " + N(jof(g.b1)) + ", " + N(jof(g.tpos_out)) + ">(M, e, c, g.cm_reg, " + lane_args + ");";
      }
      if (g.omask) L("      if ((base & g.omask) == g.omask) { " + call + " }");
      else L("      { " + call + " }");
      if (pin && (g.omask || call.compare(0, 9, "if ((base") == 0)) {
        std::string pl = "     ";
        for (int i = 0; i < 32; ++i) pl += " asm volatile(\"\" : \"+v\"(e[" + N(i) + "].x), \"+v\"(e[" + N(i) + "].y));";
        L(pl);
      }
      L("      __builtin_amdgcn_sched_barrier(0);");
      L("    }");
    }
    L("  }");
  }
  const std::string el = "e" + N(plan.passes.size() - 1);
  if (fold && fold->g) {
    L("  const uint64_t dlane = dst_of(lane_off);");
    for (int u = 0; u < 32; ++u) L("  stg<NT>(out + dst_of(wbase | " + ub(u) + ") + dlane, " + el + "[" + N(u) + "]);");
  } else {
    for (int u = 0; u < 32; ++u) L("  stg<NT>(st + (wbase | " + ub(u) + ") + lane_off, " + el + "[" + N(u) + "]);");
  }
  L("}");
  return o;
}

// Look the segment's kernel up (compile it on a miss) and — unless `launch` is null (compile-only pass before a graph capture)
// — launch it, all under the cache's mutex: between "here is the function" and "it is enqueued" no other thread may evict and
// unload it.  (Launches are asynchronous: the critical section is microseconds on a hit.)  Nothing is evicted while this
// handle records a graph: unloading synchronises the device, which a capture forbids; the cache overshoots its bound until the
// next ordinary call.
static std::string jit_key(const qip_hip_state* s, const std::string& src, bool fma) {
  return std::to_string(s->device) + (fma ? " fma\n" : "\n") + src;
}
// under g_jit_mutex: a compiled code object becomes a resident kernel of the cache
static int jit_insert_locked(qip_hip_state* s, const std::string& key, const std::vector<char>& code, hipFunction_t* fn) {
  JitKernel k;
  HIPCHK(hipModuleLoadData(&k.module, code.data()));
  hipError_t e = hipModuleGetFunction(&k.fn, k.module, "qip_segment");
  if (e != hipSuccess) {
    (void)hipModuleUnload(k.module);
    return fail(QIP_ERR_DEVICE, "hipModuleGetFunction failed: %s", hipGetErrorString(e));
  }
  k.device = s->device;
  k.last_use = ++g_jit_clock;
  g_jit_loaded += 1;
  g_jit_cache[key] = k;
  if (fn) *fn = k.fn;
  return QIP_OK;
}
// under g_jit_mutex: the code object of (src, fma) from the disk cache, or compiled here (and then stored there)
static int jit_code_locked(const std::string& src, bool fma, std::vector<char>* code) {
  const std::string dir = g_jit_disk ? jit_dir_locked() : std::string();
  JitHash h;
  std::string path;
  if (!dir.empty()) {
    h = jit_hash(src, fma);
    path = jit_disk_path(dir, h);
    const auto t0 = std::chrono::steady_clock::now();
    if (jit_disk_read(path, src.size(), h, code)) {
      g_jit_disk_hits += 1;
      g_jit_load_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      return QIP_OK;
    }
  }
  const auto t0 = std::chrono::steady_clock::now();
  QCHK(hiprtc_compile(src, fma, code));
  g_jit_compile_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  g_jit_compiles += 1;
  if (!dir.empty() && jit_disk_write(path, src.size(), h, *code)) {
    g_jit_disk_stores += 1;
    jit_disk_trim(dir);
  }
  return QIP_OK;
}
static int jit_get_and_launch(qip_hip_state* s, const std::string& src, bool fma, const std::function<int(hipFunction_t)>& launch) {
  const std::string key = jit_key(s, src, fma);
  std::lock_guard<std::mutex> lock(g_jit_mutex);
  hipFunction_t fn = nullptr;
  auto it = g_jit_cache.find(key);
  if (it != g_jit_cache.end()) {
    it->second.last_use = ++g_jit_clock;
    fn = it->second.fn;
  } else if (!launch && s->jit_collect) {
    s->jit_collect->push_back({src, fma});  // (the pre-compilation of a plan: jit_compile_collected)
    return QIP_OK;
  } else {
    std::vector<char> code;
    QCHK(jit_code_locked(src, fma, &code));
    QCHK(jit_insert_locked(s, key, code, &fn));
    if (!s->capture_pool && g_jit_captures_in_progress == 0) jit_evict_locked();  // (never the entry just inserted: it is the most recently used)
  }
  return launch ? launch(fn) : QIP_OK;
}

// The segments of a plan that are not compiled yet, compiled side by side: hiprtc programs are independent objects, one host
// thread each (at most `jit_threads`, global option); loading the code objects and entering them in the cache happens on the
// calling thread under the cache's mutex.  A compilation that fails is reported like a serial one.
// MEASURED (r4, ROCm 7.2, 8 host cores): no gain — 18 narrow segments 8.4 s on 8 threads against 9.0 s on one, 12 wide ones
// 14.3 against 15.3: the code-object manager behind hiprtc serialises the compilations of one process.  Default 1 = off
// (segments compile on first use, as before); what does make a second process fast is comgr's own on-disk cache.
int64_t g_jit_threads = 1;
static int hiprtc_compile_many(const std::vector<std::pair<std::string, bool>>& jobs, std::vector<std::vector<char>>* code_out) {
  QCHK(hiprtc_load());
  const size_t nj = jobs.size();
  std::vector<std::vector<char>>& code = *code_out;
  code.assign(nj, std::vector<char>());
  std::vector<int> rc(nj, QIP_OK);
  std::vector<std::string> msg(nj);
  const size_t nthreads = std::max<size_t>(1, std::min<size_t>({nj, (size_t)std::max<int64_t>(g_jit_threads, 1), (size_t)std::max(1u, std::thread::hardware_concurrency())}));
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    for (size_t i = next.fetch_add(1); i < nj; i = next.fetch_add(1)) {
      rc[i] = hiprtc_compile(jobs[i].first, jobs[i].second, &code[i]);
      if (rc[i] != QIP_OK) msg[i] = g_last_error;  // (thread-local: carried over to the caller's thread below)
    }
  };
  if (nthreads == 1) {
    worker();
  } else {
    std::vector<std::thread> pool;
    for (size_t t = 0; t < nthreads; ++t) pool.emplace_back(worker);
    for (auto& th : pool) th.join();
  }
  for (size_t i = 0; i < nj; ++i)
    if (rc[i] != QIP_OK) return fail(rc[i], "%s", msg[i].c_str());
  return QIP_OK;
}
// Device-free half: the code objects of `jobs` — from the disk cache, from helper processes, or compiled here.
static int jit_obtain_code(const std::vector<std::pair<std::string, bool>>& jobs, std::vector<std::vector<char>>* code_out) {
  const size_t nj = jobs.size();
  if (const char* dump = getenv("QIP_HIP_JIT_DUMP_DIR")) {  // debugging aid (tools/jit_segment_resources.py): every segment's source, numbered
    static std::atomic<unsigned> serial{0};
    if (*dump && mkdir_p(dump))
      for (size_t i = 0; i < nj; ++i) {
        char name[64];
        snprintf(name, sizeof name, "/seg_%04u%s.hip", serial.fetch_add(1), jobs[i].second ? "_fma" : "");
        if (FILE* f = fopen((std::string(dump) + name).c_str(), "wb")) {
          (void)fwrite(jobs[i].first.data(), 1, jobs[i].first.size(), f);
          fclose(f);
        }
      }
  }
  std::vector<std::vector<char>>& code = *code_out;
  code.assign(nj, std::vector<char>());
  std::string dir, helper;
  int procs = 1;
  {
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    if (g_jit_disk) dir = jit_dir_locked();
    procs = jit_procs_effective();
  }
  // 1. what the disk cache already holds
  std::vector<JitHash> hashes(nj);
  std::vector<std::string> paths(nj);
  std::vector<size_t> todo;
  double load_ms = 0;
  uint64_t hits = 0;
  for (size_t i = 0; i < nj; ++i) {
    if (!dir.empty()) {
      hashes[i] = jit_hash(jobs[i].first, jobs[i].second);
      paths[i] = jit_disk_path(dir, hashes[i]);
      const auto t0 = std::chrono::steady_clock::now();
      if (jit_disk_read(paths[i], jobs[i].first.size(), hashes[i], &code[i])) {
        load_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        hits += 1;
        continue;
      }
    }
    todo.push_back(i);
  }
  // 2. the rest: side by side in helper processes when there are several (and somewhere to put the results) ...
  const auto t0 = std::chrono::steady_clock::now();
  uint64_t by_helpers = 0;
  // (a process with a foreign libhiprtc sends single segments out too — unless "jit_procs" = 1 asks for this process only)
  if (!todo.empty() && !dir.empty() && ((todo.size() >= 2 && procs > 1) || (g_jit_procs != 1 && jit_foreign_hiprtc_loaded()))) {
    helper = jit_helper_path();
    if (!helper.empty()) {
      jit_compile_in_helpers(helper, dir, procs, jobs, todo, paths);
      std::vector<size_t> still;
      for (size_t i : todo) {
        if (jit_disk_read(paths[i], jobs[i].first.size(), hashes[i], &code[i])) by_helpers += 1;
        else still.push_back(i);
      }
      todo.swap(still);
    }
  }
  // 3. ... and whatever is left (no helper, a single segment, a helper that failed: the message comes from here) in this process
  uint64_t compiled_here = 0, stored = 0;
  if (!todo.empty()) {
    std::vector<std::pair<std::string, bool>> rest;
    for (size_t i : todo) rest.push_back(jobs[i]);
    std::vector<std::vector<char>> rest_code;
    QCHK(hiprtc_compile_many(rest, &rest_code));
    for (size_t j = 0; j < todo.size(); ++j) {
      code[todo[j]].swap(rest_code[j]);
      compiled_here += 1;
      if (!dir.empty() && jit_disk_write(paths[todo[j]], jobs[todo[j]].first.size(), hashes[todo[j]], code[todo[j]])) stored += 1;
    }
  }
  const double compile_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::lock_guard<std::mutex> lock(g_jit_mutex);
  g_jit_disk_hits += hits;
  g_jit_load_ms += load_ms;
  g_jit_helper_segments += by_helpers;
  g_jit_compiles += compiled_here + by_helpers;
  g_jit_disk_stores += stored + by_helpers;
  if (compiled_here + by_helpers) g_jit_compile_ms += compile_ms;
  if (stored + by_helpers) jit_disk_trim(dir);
  return QIP_OK;
}

// ---- r6: compiled sweeps for one-shot callers, when they are free (option "tile_auto") -------------------------------------
// apply_ops on a state with tile >= 1 and tile_jit = 0 is what a `calculate_state` caller issues (HipBuilder): once.  It does not
// repay seconds of compilation — but a plan whose segments are ALL already resident or in the disk cache costs milliseconds to
// load and runs 1.5x faster than the interpreter.  So such a call LOOKS its wide plan up (jit_load_only): all hits -> compiled
// sweeps (bit-identical to the interpreter for tile = 1); any miss -> the interpreter runs now, and the missing segments are
// handed to helper processes nobody waits for (jit_compile_in_background): the next call, or the next process, finds them.
static constexpr int kJitMiss = -1001;  // internal status of a lookup-only pass (never crosses the ABI)
static std::map<std::string, std::chrono::steady_clock::time_point> g_jit_inflight;  // object paths handed to background helpers
static std::map<uint64_t, std::chrono::steady_clock::time_point> g_jit_cold_plans;   // plan fingerprints last seen incomplete

// the jobs that are resident or on disk become resident; the others are returned in `misses` (nothing is compiled)
static int jit_load_only(qip_hip_state* s, std::vector<std::pair<std::string, bool>>& jobs, std::vector<std::pair<std::string, bool>>* misses) {
  {
    std::vector<std::pair<std::string, bool>> uniq;
    for (auto& j : jobs)
      if (std::find(uniq.begin(), uniq.end(), j) == uniq.end()) uniq.push_back(std::move(j));
    jobs.swap(uniq);
  }
  std::string dir;
  {
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    if (g_jit_disk) dir = jit_dir_locked();
  }
  std::vector<std::vector<char>> code(jobs.size());
  std::vector<char> hit(jobs.size(), 0);
  uint64_t hits = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (size_t i = 0; i < jobs.size() && !dir.empty(); ++i) {
    const JitHash h = jit_hash(jobs[i].first, jobs[i].second);
    if (jit_disk_read(jit_disk_path(dir, h), jobs[i].first.size(), h, &code[i])) {
      hit[i] = 1;
      hits += 1;
    }
  }
  const double load_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  for (size_t i = 0; i < jobs.size(); ++i)
    if (!hit[i]) misses->push_back(jobs[i]);
  if (!misses->empty()) return QIP_OK;  // (a partial plan is of no use: nothing is loaded onto the device)
  std::lock_guard<std::mutex> lock(g_jit_mutex);
  g_jit_disk_hits += hits;
  g_jit_load_ms += load_ms;
  for (size_t i = 0; i < jobs.size(); ++i) {
    const std::string key = jit_key(s, jobs[i].first, jobs[i].second);
    if (g_jit_cache.find(key) != g_jit_cache.end()) continue;
    QCHK(jit_insert_locked(s, key, code[i], nullptr));
  }
  if (!s->capture_pool && g_jit_captures_in_progress == 0) jit_evict_locked();
  return QIP_OK;
}

// Hand `jobs` to helper processes and return at once.  The helpers write the code objects into the disk cache and remove their
// own source files (qip_jitc -u); a detached thread reaps them (it touches nothing of this library).  Jobs handed out less than
// two minutes ago are not handed out again.
static void jit_compile_in_background(const std::vector<std::pair<std::string, bool>>& jobs) {
  std::string dir, helper;
  int procs = 1;
  std::vector<size_t> todo;
  std::vector<std::string> out_paths(jobs.size());
  {
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    if (g_jit_disk) dir = jit_dir_locked();
    if (dir.empty()) return;  // nowhere to leave the results
    procs = jit_procs_effective();
    const auto now = std::chrono::steady_clock::now();
    for (auto it = g_jit_inflight.begin(); it != g_jit_inflight.end();)
      it = now - it->second > std::chrono::seconds(120) ? g_jit_inflight.erase(it) : std::next(it);
    for (size_t i = 0; i < jobs.size(); ++i) {
      out_paths[i] = jit_disk_path(dir, jit_hash(jobs[i].first, jobs[i].second));
      if (g_jit_inflight.count(out_paths[i])) continue;
      g_jit_inflight[out_paths[i]] = now;
      todo.push_back(i);
    }
    g_jit_background_segments += todo.size();
  }
  if (todo.empty()) return;
  helper = jit_helper_path();
  if (helper.empty()) return;
  std::vector<std::string> src_paths(jobs.size());
  std::vector<size_t> written;
  for (size_t i : todo) {
    char name[64];
    snprintf(name, sizeof name, "/seg.%ld.%u.%zu.hip", (long)getpid(), g_jit_tmp_serial.fetch_add(1), i);
    src_paths[i] = dir + name;
    const int fd = open(src_paths[i].c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) continue;
    const bool ok = write(fd, jobs[i].first.data(), jobs[i].first.size()) == (ssize_t)jobs[i].first.size();
    if (close(fd) == 0 && ok) written.push_back(i);
    else (void)unlink(src_paths[i].c_str());
  }
  // (half of the usual share: the caller's own sweeps are running on this host's cores' attention too)
  const size_t np = std::min<size_t>((size_t)std::max(1, procs / 2), written.size());
  std::vector<pid_t> pids;
  for (size_t k = 0; k < np; ++k) {
    std::vector<std::string> args = {helper, "-u"};
    for (size_t j = k; j < written.size(); j += np) {
      const size_t i = written[j];
      args.push_back(jobs[i].second ? "1" : "0");
      args.push_back(src_paths[i]);
      args.push_back(out_paths[i]);
    }
    std::vector<char*> argv;
    for (std::string& a : args) argv.push_back(&a[0]);
    argv.push_back(nullptr);
    pid_t pid = 0;
    if (posix_spawn(&pid, helper.c_str(), nullptr, nullptr, argv.data(), environ) == 0) pids.push_back(pid);
  }
  {
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    g_jit_helper_procs += pids.size();
  }
  if (pids.empty()) {
    for (size_t i : written) (void)unlink(src_paths[i].c_str());
    return;
  }
  std::thread([pids]() {
    for (pid_t pid : pids) {
      int status = 0;
      while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {
      }
    }
  }).detach();
}
static int jit_compile_collected(qip_hip_state* s, std::vector<std::pair<std::string, bool>>& jobs) {
  {  // the same segment twice in one plan: once
    std::vector<std::pair<std::string, bool>> uniq;
    for (auto& j : jobs)
      if (std::find(uniq.begin(), uniq.end(), j) == uniq.end()) uniq.push_back(std::move(j));
    jobs.swap(uniq);
  }
  if (jobs.empty()) return QIP_OK;
  std::vector<std::vector<char>> code;
  QCHK(jit_obtain_code(jobs, &code));
  std::lock_guard<std::mutex> lock(g_jit_mutex);
  for (size_t i = 0; i < jobs.size(); ++i) {
    const std::string key = jit_key(s, jobs[i].first, jobs[i].second);
    if (g_jit_cache.find(key) != g_jit_cache.end()) continue;  // (another thread's handle got there first)
    QCHK(jit_insert_locked(s, key, code[i], nullptr));
  }
  if (!s->capture_pool && g_jit_captures_in_progress == 0) jit_evict_locked();
  return QIP_OK;
}

template <typename T>
static int launch_tile_segment(qip_hip_state* s, const std::vector<const TileItem*>& seg,
                               std::vector<uint32_t> high_in, uint32_t p5_override = 0,
                               const std::vector<uint32_t>* grid_ctl = nullptr, double alg_bytes = 0,
                               const TileStorePerm* fold = nullptr) {
  // `fold`: store the tiles into the second buffer, packed for the multi-GPU exchange (TileStorePerm), and make that buffer
  // current — the remap's gather rides in this sweep's store phase instead of costing a sweep of its own
  // `grid_ctl` (one-op sweeps): control positions OUTSIDE the tile that are taken off the grid — only the blocks whose base
  // reads 1 there are launched, so a controlled gate sweeps half / a quarter of the vector like the dedicated kernels do
  // (the kernel's own test of `omask` against the block's base then always passes).  `alg_bytes`: what the profile credits.
  arena_begin_group(s);
  TileSegmentPlan<T> plan;
  QCHK(build_tile_segment<T>(s->n, s->tile_passes != 0, seg, std::move(high_in), &plan, s->tile >= 2 ? 2 : 1, p5_override));
  const std::vector<uint32_t>& high = plan.high;
  // r5: the interpreter kernel takes runs of diagonal gates as one loop over TileDiagItem steps (tile_merge_diag_runs: the same
  // products in the same order, without the per-gate decoding); the plan itself stays what the generators and the CPU replay read
  TileInterpPlan<T> interp;
  const bool use_runs = g_tile_diag_runs && s->tile_passes && !s->tile_jit;
  if (use_runs) tile_merge_diag_runs<T>(plan, &interp);
  const bool has_runs = use_runs && interp.runs > 0;
  std::vector<TileGate<T>>& gates = has_runs ? interp.gates : plan.gates;
  std::vector<amp_t<T>>& mats = plan.mats;
  TilePassDesc& pd = has_runs ? interp.pd : plan.pd;
  const size_t gates_bytes = gates.size() * sizeof(TileGate<T>);
  const size_t mats_bytes = mats.size() * sizeof(amp_t<T>);
  const size_t items_bytes = has_runs ? interp.items.size() * sizeof(TileDiagItem<T>) : 0;
  static_assert(sizeof(TileGate<T>) % 16 == 0, "the matrix block behind the gate list stays 16-byte aligned");
  static_assert(sizeof(amp_t<T>) * 16 % 16 == 0 && sizeof(TileDiagItem<T>) % 16 == 0, "... and the diagonal steps behind the matrices");
  const size_t items_off = (gates_bytes + mats_bytes + 15) / 16 * 16;
  auto upload_gates = [&]() -> int {  // after the gates are final (k_tile_passes resolves them per pass first)
    QCHK(ensure_arena(s, items_off + items_bytes));  // one allocation: growing frees the old arena
    QCHK(arena_upload(s, gates.data(), gates_bytes, 0));
    if (!mats.empty()) QCHK(arena_upload(s, mats.data(), mats_bytes, gates_bytes));
    if (items_bytes) QCHK(arena_upload(s, interp.items.data(), items_bytes, items_off));
    return QIP_OK;
  };
  TileDesc d;
  memset(&d, 0, sizeof d);
  d.ngates = (uint32_t)gates.size();
  for (int j = 0; j < kTileHigh; ++j) d.hpos[j] = high[j];
  d.p5 = plan.p5;
  Ins ins = tile_ins(high, plan.p5);  // (sorts its own copy; `high` keeps the tile-bit order)
  uint64_t ntiles = 1ull << (s->n - kTileBits);
  if (grid_ctl && !grid_ctl->empty()) {
    std::vector<uint32_t> opened = high, ctl = *grid_ctl;
    for (uint32_t c : ctl) opened.push_back(c);
    for (uint32_t& o : opened)
      if (o == 5u) o = plan.p5;  // (tile_block_base: the space where p5 and 5 have traded places)
    uint64_t ones = 0;
    for (uint32_t c : ctl) ones |= 1ull << (c == 5u ? plan.p5 : c);
    ins = make_ins(opened, ones);
    ntiles >>= ctl.size();
  }
  const double sweep_bytes = alg_bytes > 0 ? alg_bytes : 2.0 * (double)s->amp_bytes * (double)s->namps;
  // a packed store needs every tile written (no blocks taken off the grid), the pass kernel, rows that stay rows (none of
  // the gathered positions is a lane position of this tile) and the second buffer; otherwise the sweep runs as usual and the
  // remap gathers by itself
  if (!fold && s->fold_now && s->fold_request && !s->fold_done) fold = s->fold_request;
  bool folding = fold && fold->g && s->tile_passes && (!grid_ctl || grid_ctl->empty()) && !s->capture_pool && !s->jit_prepare;
  if (folding)
    for (uint32_t t = 0; t < fold->g; ++t) folding = folding && !tile_is_low(fold->sel[t], plan.p5);
  if (folding) QCHK(ensure_alt(s));
  // r5: the sweep in 2^nbits parts (TileSlicing, qip_internal.h) — the slice positions must be block-index bits of this sweep
  // (not tile positions, not positions taken off the grid), above the rows, and with `need_fold` the store must really be packed
  TileSlicing* sl = (s->capture_pool || s->jit_prepare) ? nullptr : s->slice_now;
  bool slicing = sl && sl->nbits >= 1 && sl->nbits <= 3 && s->tile_passes && (!grid_ctl || grid_ctl->empty()) && (!sl->need_fold || folding) &&
                 !(sl->in_place_only && folding) && s->n >= (uint32_t)kTileBits + sl->nbits;
  if (slicing)
    for (uint32_t j = 0; j < sl->nbits; ++j) {
      slicing = slicing && sl->pos[j] > 11u && sl->pos[j] < s->n && !tile_is_low(sl->pos[j], plan.p5) &&
                std::find(high.begin(), high.end(), sl->pos[j]) == high.end();
      for (uint32_t i = 0; i < j; ++i) slicing = slicing && sl->pos[i] != sl->pos[j];
    }
  Ins ins_sliced = ins;
  if (slicing) {
    std::vector<uint32_t> opened = high;
    for (uint32_t& o : opened)
      if (o == 5u) o = plan.p5;  // (as above: the space where p5 and 5 have traded places; slice positions are > 11)
    for (uint32_t j = 0; j < sl->nbits; ++j) opened.push_back(sl->pos[j]);
    ins_sliced = make_ins(opened, 0);
  }
  const uint32_t nparts = slicing ? 1u << sl->nbits : 1u;
  auto slice_or = [&](uint32_t k) {
    uint64_t v = 0;
    for (uint32_t j = 0; slicing && j < sl->nbits; ++j) v |= (uint64_t)((k >> j) & 1u) << sl->pos[j];
    return v;
  };
  if (sl && !slicing && sl->fallback) QCHK(sl->fallback());
  if (sl) s->slice_now = nullptr;  // (consumed: by this step, sliced or not)
  auto folded_swap = [&]() {
    std::swap(s->cur, s->alt);
    std::swap(s->owns_cur, s->owns_alt);
    s->fold_done = true;
  };
  const size_t lds = sizeof(amp_t<T>) << kTileBits;
  const TileGate<T>* dg = nullptr;  // device addresses: valid only after the upload (the arena may grow / move)
  const amp_t<T>* dmats = nullptr;
  const TileDiagItem<T>* ditems = nullptr;
  ProfRec rec;
  rec.cls = KC_TILE_GATES;
  auto begin = [&]() -> int {  // descriptors up, then the timed region starts
    QCHK(upload_gates());
    dg = (const TileGate<T>*)s->arena;
    dmats = (const amp_t<T>*)((const char*)s->arena + gates_bytes);
    ditems = (const TileDiagItem<T>*)((const char*)s->arena + items_off);
    if (s->profile) QCHK(prof_begin(s, KC_TILE_GATES, sweep_bytes, &rec));
    return QIP_OK;
  };
  if (s->tile_passes && s->tile_jit) {
    // the segment as its own kernel: nothing to upload, the descriptors are constants of the code
    const bool fma = s->tile_fma && s->tile >= 2;  // tile = 1 promises IEEE equality with the gate-by-gate path: never fused
    // structure compiled, numbers in the arena (see tile_jit_source): the default form.  tile_jit = 3 (tuning aid) writes the
    // numbers into the source as literals instead: every constant then occupies vector registers (gfx950's VOP3 takes no
    // 64-bit literal), which makes the configs[1] segments spill under the five-blocks-per-CU register bound (104 vs 90 ms),
    // and every new angle is a new kernel; its one merit is ~3 % fewer multiplies in QFT where equal constants fold
    const bool parametrised = s->tile_jit != 3;
    std::vector<T> params;
    const bool merge = s->tile_merge && s->tile >= 2;  // products of runs of diagonal gates: rounding differs (1e-12 mode only)
    int remap = ntiles % 64 == 0 ? (int)g_tile_remap : 0;
    if (remap == 4) {  // XCD-aware: tile-number bit J is the first free position at or above 16 (see tile_jit_source)
      uint32_t nb = 0, J = 0;
      while ((1ull << nb) < ntiles) ++nb;
      uint32_t seen = 0;
      for (uint32_t pp = kTileLow; pp < s->n && seen < nb; ++pp) {
        bool opened = false;
        for (uint32_t j = 0; j < ins.npos; ++j) opened = opened || ins.pos[j] == pp;
        if (opened) continue;
        if (pp < 16) ++J;
        ++seen;
      }
      if (nb < 3) remap = 0;
      else remap = 16 + (int)std::min(J, nb - 3);
    }
    if (slicing) remap = 0;
    const std::string src = tile_jit_source<T>(plan, slicing ? ins_sliced : ins, use_nt(s), remap, parametrised ? &params : nullptr, merge,
                                               folding ? fold : nullptr, slicing);
    QCHK(jit_get_and_launch(s, src, fma, nullptr));  // compile on a miss BEFORE the timed region starts
    if (s->jit_prepare) return QIP_OK;
    if (parametrised && !params.empty()) QCHK(arena_upload(s, params.data(), params.size() * sizeof(T), 0));
    if (s->profile) QCHK(prof_begin(s, KC_TILE_GATES, sweep_bytes, &rec));
    void* st_ptr = s->cur;
    uint64_t ntiles_arg = ntiles;  // (a sliced kernel reads the part's slice bits here instead)
    void* params_ptr = s->arena;
    void* out_ptr = s->alt;
    // (a kernel without parameters has no third argument: the packed-store destination then comes third)
    void* args_p[] = {&st_ptr, &ntiles_arg, &params_ptr, &out_ptr};
    void* args_np[] = {&st_ptr, &ntiles_arg, &out_ptr};
    void** args = (parametrised || !folding) ? args_p : args_np;
    const dim3 grid = grid2d(ntiles >> (slicing ? sl->nbits : 0), 1);
    for (uint32_t k = 0; k < nparts; ++k) {
      if (slicing && sl->before) QCHK(sl->before(k, folding));
      if (slicing) ntiles_arg = slice_or(k);
      QCHK(jit_get_and_launch(s, src, fma, [&](hipFunction_t fn) -> int {
        HIPCHK(hipModuleLaunchKernel(fn, grid.x, grid.y, 1, kTileBlock, 1, 1, (unsigned)lds, s->stream, args, nullptr));
        return QIP_OK;
      }));
      if (slicing && sl->after) QCHK(sl->after(k, folding));
      if (slicing) sl->parts_done += 1;
      if (slicing && s->profile) s->prof_launches[KC_TILE_PARTS] += 1;
    }
    if (slicing) sl->folded = folding;
    if (s->profile) QCHK(prof_end(s, &rec));
    if (folding) folded_swap();
    return QIP_OK;
  }
  if (s->jit_prepare) return QIP_OK;
  if (s->tile_passes) {
    QCHK(begin());
    const uint64_t ntiles_part = ntiles >> (slicing ? sl->nbits : 0);
    for (uint32_t k = 0; k < nparts; ++k) {
      Ins ins_k = slicing ? ins_sliced : ins;
      ins_k.ormask |= slice_or(k);
      if (slicing && sl->before) QCHK(sl->before(k, folding));
      if (folding) {
#define TPF(NTV) hipLaunchKernelGGL((k_tile_passes<T, NTV, true>), grid2d(ntiles_part, 1), dim3(kTileBlock), lds, s->stream, \
                                    (amp_t<T>*)s->cur, ins_k, pd, dg, dmats, (amp_t<T>*)s->alt, *fold, ditems)
        if (use_nt(s)) TPF(true);
        else TPF(false);
#undef TPF
      } else {
#define TP(NTV) hipLaunchKernelGGL((k_tile_passes<T, NTV>), grid2d(ntiles_part, 1), dim3(kTileBlock), lds, s->stream, \
                                   (amp_t<T>*)s->cur, ins_k, pd, dg, dmats, (amp_t<T>*)nullptr, TileStorePerm(), ditems)
        if (use_nt(s)) TP(true);
        else TP(false);
#undef TP
      }
      HIPCHK(hipGetLastError());
      if (slicing && sl->after) QCHK(sl->after(k, folding));
      if (slicing) sl->parts_done += 1;
      if (slicing && s->profile) s->prof_launches[KC_TILE_PARTS] += 1;
    }
    if (slicing) sl->folded = folding;
    if (folding) {
      if (s->profile) QCHK(prof_end(s, &rec));
      folded_swap();
      return QIP_OK;
    }
  } else {
    QCHK(begin());
    if (use_nt(s))
      hipLaunchKernelGGL((k_tile_gates<T, true>), grid2d(ntiles, 1), dim3(kBlock), lds, s->stream,
                         (amp_t<T>*)s->cur, ins, d, dg);
    else
      hipLaunchKernelGGL((k_tile_gates<T, false>), grid2d(ntiles, 1), dim3(kBlock), lds, s->stream,
                         (amp_t<T>*)s->cur, ins, d, dg);
  }
  HIPCHK(hipGetLastError());
  if (s->profile) QCHK(prof_end(s, &rec));
  return QIP_OK;
}

// a multi-gate step of a wide plan (option "tile_wide"): always its own run-time-compiled kernel
template <typename T>
static int launch_wide_segment(qip_hip_state* s, const std::vector<const TileItem*>& seg, std::vector<uint32_t> high_in) {
  arena_begin_group(s);
  WidePlan<T> plan;
  QCHK(build_wide_segment<T>(s->n, seg, std::move(high_in), &plan, s->tile >= 2 ? 2 : 1));
  const Ins ins = tile_ins(plan.high, plan.p5);
  const bool fma = s->tile_fma && s->tile >= 2;
  const bool parametrised = s->tile_jit != 3;
  std::vector<T> params;
  const bool merge = s->tile_merge && s->tile >= 2;  // products of runs of diagonal gates: rounding differs (1e-12 mode only)
  // r5: a packed store (the multi-GPU remap's gather rides in this sweep, TileStorePerm) under the same conditions as the 11-bit
  // sweeps: none of the gathered positions is a lane position of the tile, the second buffer exists
  const TileStorePerm* fold = (s->fold_now && s->fold_request && !s->fold_done) ? s->fold_request : nullptr;
  bool folding = fold && fold->g && !s->capture_pool && !s->jit_prepare;
  if (folding)
    for (uint32_t t = 0; t < fold->g; ++t) folding = folding && !tile_is_low(fold->sel[t], plan.p5);
  if (folding) QCHK(ensure_alt(s));
  // r5: the sweep in parts (TileSlicing: the sharded state's exchange overlaps with it)
  TileSlicing* sl = (s->capture_pool || s->jit_prepare) ? nullptr : s->slice_now;
  bool slicing = sl && sl->nbits >= 1 && sl->nbits <= 3 && (!sl->need_fold || folding) && !(sl->in_place_only && folding) &&
                 s->n >= (uint32_t)kWideBits + sl->nbits;
  if (slicing)
    for (uint32_t j = 0; j < sl->nbits; ++j) {
      slicing = slicing && sl->pos[j] > 11u && sl->pos[j] < s->n && !tile_is_low(sl->pos[j], plan.p5) &&
                std::find(plan.high.begin(), plan.high.end(), sl->pos[j]) == plan.high.end();
      for (uint32_t i = 0; i < j; ++i) slicing = slicing && sl->pos[i] != sl->pos[j];
    }
  Ins ins_use = ins;
  if (slicing) {
    std::vector<uint32_t> opened = plan.high;
    for (uint32_t& o : opened)
      if (o == 5u) o = plan.p5;
    for (uint32_t j = 0; j < sl->nbits; ++j) opened.push_back(sl->pos[j]);
    ins_use = make_ins(opened, 0);
  }
  if (sl && !slicing && sl->fallback) QCHK(sl->fallback());
  if (sl) s->slice_now = nullptr;
  const std::string src = wide_jit_source<T>(plan, ins_use, use_nt(s), parametrised ? &params : nullptr, merge, g_tile_wide_pin != 0, g_tile_wide_dense3_inline != 0, slicing,
                                             folding ? fold : nullptr);
  if (src.empty()) return fail(QIP_ERR_INVALID, "internal: wide segment source");
  QCHK(jit_get_and_launch(s, src, fma, nullptr));  // compile on a miss before the timed region starts
  if (s->jit_prepare) return QIP_OK;
  if (parametrised && !params.empty()) QCHK(arena_upload(s, params.data(), params.size() * sizeof(T), 0));
  ProfRec rec;
  rec.cls = KC_TILE_GATES;
  if (s->profile) QCHK(prof_begin(s, KC_TILE_GATES, 2.0 * (double)s->amp_bytes * (double)s->namps, &rec));
  void* st_ptr = s->cur;
  const uint64_t ntiles = (1ull << (s->n - (uint32_t)kWideBits)) >> (slicing ? sl->nbits : 0);
  uint64_t second_arg = ntiles;  // (a sliced kernel reads the part's slice bits here)
  void* params_ptr = s->arena;
  void* out_ptr = s->alt;
  // (a kernel without parameters has no third argument: the packed-store destination then comes third)
  void* args_p[] = {&st_ptr, &second_arg, &params_ptr, &out_ptr};
  void* args_np[] = {&st_ptr, &second_arg, &out_ptr};
  void** args = (parametrised || !folding) ? args_p : args_np;
  const dim3 grid = grid2d(ntiles, 1);
  const size_t lds = 2 * (sizeof(amp_t<T>) << kTileBits);  // the transposition buffer: two quarters of the tile
  const uint32_t nparts = slicing ? 1u << sl->nbits : 1u;
  for (uint32_t k = 0; k < nparts; ++k) {
    if (slicing) {
      second_arg = 0;
      for (uint32_t j = 0; j < sl->nbits; ++j) second_arg |= (uint64_t)((k >> j) & 1u) << sl->pos[j];
      if (sl->before) QCHK(sl->before(k, folding));
    }
    QCHK(jit_get_and_launch(s, src, fma, [&](hipFunction_t fn) -> int {
      HIPCHK(hipModuleLaunchKernel(fn, grid.x, grid.y, 1, 256, 1, 1, (unsigned)lds, s->stream, args, nullptr));
      return QIP_OK;
    }));
    if (slicing) {
      if (sl->after) QCHK(sl->after(k, folding));
      sl->parts_done += 1;
      if (s->profile) s->prof_launches[KC_TILE_PARTS] += 1;
    }
  }
  if (slicing) sl->folded = folding;
  if (s->profile) QCHK(prof_end(s, &rec));
  if (folding) {
    std::swap(s->cur, s->alt);
    std::swap(s->owns_cur, s->owns_alt);
    s->fold_done = true;
  }
  return QIP_OK;
}

template <typename T>
int tile_apply_single(qip_hip_state* s, const qip_op* op, bool* done, double alg_bytes) {
  *done = false;
  if (s->n < (uint32_t)kTileBits || !s->tile_passes) return QIP_OK;
  TileItem it;
  QCHK(classify_tile_item(s->dtype, s->n, op, &it));
  std::vector<TileItem> parts;
  if (it.tileable) {
    parts.push_back(it);
  } else if (it.swap_pairs.size() >= 2) {
    // an uncontrolled Swap(h >= 2) is h disjoint transpositions: h bit-swap items of ONE sweep (pure moves: any grouping is
    // bit-identical to the single permutation)
    for (const auto& pr : it.swap_pairs) {
      TileItem t;
      t.tileable = t.exact = true;
      t.kind = 2;
      t.t0 = std::min(pr.first, pr.second);
      t.t1 = std::max(pr.first, pr.second);
      t.pos = {t.t0, t.t1};
      t.nd_mask = (1ull << t.t0) | (1ull << t.t1);
      parts.push_back(t);
    }
  } else {
    return QIP_OK;
  }
  // Which rows for this ONE op (qip_kernels.h tile_block_base)?  With at most three positions forced above contiguous rows the
  // two free ones become 11 and 12 — the fastest tile shape measured (5.3 ms per sweep at n = 30) — so contiguous rows stay;
  // with more, the forced positions scatter the tile over the DRAM row bits and the split rows win (k = 4 on four high
  // positions: 72.9 -> 75.5 %, profiles/r04_tile_rows.md).
  // Only the positions the op EXCHANGES amplitudes across must be tile bits; its controls may sit anywhere: inside the rows they
  // are lane predicates, above them they come off the grid (r4: controlled dense k = 2, 3 take this route too).
  auto exchanged = [](const TileItem& t) {
    std::vector<uint32_t> e;
    for (uint32_t p : t.pos)
      if ((t.nd_mask >> p) & 1ull) e.push_back(p);
    return e;
  };
  uint32_t p5 = tile_p5(s->dtype, s->n);
  if (p5 != 5u) {
    std::vector<uint32_t> forced;
    for (const TileItem& t : parts)
      for (uint32_t p : exchanged(t))
        if (p >= (uint32_t)kTileLow && std::find(forced.begin(), forced.end(), p) == forced.end()) forced.push_back(p);
    if (forced.size() <= 3) p5 = 5u;
  }
  std::vector<uint32_t> hp;
  for (const TileItem& t : parts)
    for (uint32_t p : exchanged(t))
      if (!tile_is_low(p, p5) && std::find(hp.begin(), hp.end(), p) == hp.end()) hp.push_back(p);
  if (hp.size() > (size_t)kTileHigh) return QIP_OK;
  std::vector<uint32_t> grid_ctl;  // (a single item: `parts` holds several only for an uncontrolled Swap)
  if (parts.size() == 1)
    for (uint32_t c : parts[0].cpos)
      if (!tile_is_low(c, p5)) grid_ctl.push_back(c);
  if (s->n < (uint32_t)kTileBits + (uint32_t)grid_ctl.size()) return QIP_OK;
  std::vector<const TileItem*> seg;
  for (const TileItem& t : parts) seg.push_back(&t);
  // the free positions that pad the tile must not be control positions that left the grid
  if (!grid_ctl.empty()) {
    const uint32_t pad_from = std::min<uint32_t>((uint32_t)g_tile_pad_from, s->n > (uint32_t)kTileBits ? s->n - 5 : (uint32_t)kTileLow);
    for (int round = 0; round < 2; ++round)
      for (uint32_t p = round == 0 ? std::max<uint32_t>(pad_from, kTileLow) : 5u; hp.size() < (size_t)kTileHigh && p < s->n; ++p)
        if (!tile_is_low(p, p5) && std::find(hp.begin(), hp.end(), p) == hp.end() && std::find(grid_ctl.begin(), grid_ctl.end(), p) == grid_ctl.end())
          hp.push_back(p);
    if (hp.size() != (size_t)kTileHigh) return QIP_OK;
  }
  const int64_t jit = s->tile_jit;
  s->tile_jit = 0;  // one op does not repay a run-time compilation: the interpreter kernel (no return between here and the restore)
  const int rc = launch_tile_segment<T>(s, seg, hp, p5, &grid_ctl, alg_bytes);
  s->tile_jit = jit;
  QCHK(rc);
  *done = true;
  return QIP_OK;
}
template int tile_apply_single<double>(qip_hip_state*, const qip_op*, bool*, double);
template int tile_apply_single<float>(qip_hip_state*, const qip_op*, bool*, double);

template <typename T>
static int debug_jit_t(int dtype, uint32_t n, const qip_op* ops, uint64_t count, int mode, uint64_t* nseg, uint64_t* src_bytes,
                       uint64_t* code_bytes, std::string* first) {
  TileSchedule sc;
  QCHK(make_tile_schedule(dtype, n, ops, count, mode, true, &sc));
  const std::vector<TileItem>& items = sc.items;
  *nseg = *src_bytes = *code_bytes = 0;
  std::vector<std::pair<std::string, bool>> jobs;
  for (const TileStep& st : sc.steps) {
    if (st.ops.size() < 2 || !st.perm.empty()) continue;
    std::vector<const TileItem*> seg;
    for (uint64_t i : st.ops) seg.push_back(&items[i]);
    std::vector<T> params;  // mode bit 6: parametrised (numbers as kernel data)
    std::string src;
    // mode bit 11 (2048): the SLICED + PACKED-STORE variant the sharded state's overlapped exchange launches (r5) — one position that is
    // not a tile position is gathered on top, another cuts the sweep in two parts: generated and compiled here, run on the GPU by
    // tests/dist_worker_gpu.py
    auto variant = [&](const std::vector<uint32_t>& high, uint32_t p5, Ins* ins, TileStorePerm* sp) {
      std::vector<uint32_t> spare;
      for (uint32_t pp = n; pp-- > 12u && spare.size() < 2;)
        if (!tile_is_low(pp, p5) && std::find(high.begin(), high.end(), pp) == high.end()) spare.push_back(pp);
      if (spare.size() < 2) return false;
      memset(sp, 0, sizeof *sp);
      sp->g = 1;
      sp->Lg = n - 1;
      sp->sel[0] = sp->sel_desc[0] = spare[0];
      std::vector<uint32_t> opened = high;
      for (uint32_t& o : opened)
        if (o == 5u) o = p5;
      opened.push_back(spare[1]);
      *ins = make_ins(opened, 0);
      return true;
    };
    if ((mode & 16) && n > (uint32_t)kWideBits) {  // mode bit 4: wide tiles
      WidePlan<T> wplan;
      QCHK(build_wide_segment<T>(n, seg, st.high, &wplan, mode & 3));
      Ins ins = tile_ins(wplan.high, wplan.p5);
      TileStorePerm sp;
      const bool var = (mode & 2048) && variant(wplan.high, wplan.p5, &ins, &sp);
      src = wide_jit_source<T>(wplan, ins, true, (mode & 64) ? &params : nullptr, (mode & 128) != 0, (mode & 256) != 0, (mode & 512) != 0, var, var ? &sp : nullptr);
      if (src.empty()) return fail(QIP_ERR_INVALID, "internal: wide segment source");
    } else {
      TileSegmentPlan<T> plan;
      QCHK(build_tile_segment<T>(n, true, seg, st.high, &plan, mode & 3));
      Ins ins = tile_ins(plan.high, plan.p5);
      TileStorePerm sp;
      const bool var = (mode & 2048) && variant(plan.high, plan.p5, &ins, &sp);
      src = tile_jit_source<T>(plan, ins, true, 0, (mode & 64) ? &params : nullptr, (mode & 128) != 0, var ? &sp : nullptr, var);  // bit 7: merged diagonal runs
    }
    if (*nseg == 0 && first) *first = src;
    *nseg += 1;
    *src_bytes += src.size();
    jobs.push_back({src, (mode & 32) != 0});  // mode bit 5: fused multiply-adds
  }
  std::vector<std::vector<char>> code;  // as apply_ops obtains them: disk cache, helper processes, this process
  QCHK(jit_obtain_code(jobs, &code));
  for (const auto& c : code) *code_bytes += c.size();
  return QIP_OK;
}

extern "C" int qip_hip_debug_tile_jit(int dtype, uint32_t n, const qip_op* ops, uint64_t count, int mode, uint64_t* segments,
                                      uint64_t* source_bytes, uint64_t* code_bytes, const char** first_source) try {
  static thread_local std::string first;
  if ((count && !ops) || !segments || !source_bytes || !code_bytes) return fail(QIP_ERR_INVALID, "null argument");
  if (dtype != QIP_C64 && dtype != QIP_C32) return fail(QIP_ERR_INVALID, "bad dtype %d", dtype);
  if (n < (uint32_t)kTileBits) return fail(QIP_ERR_UNSUPPORTED, "tile sweeps need n >= %d", kTileBits);
  first.clear();
  QCHK(dtype == QIP_C64 ? debug_jit_t<double>(dtype, n, ops, count, mode, segments, source_bytes, code_bytes, &first)
                        : debug_jit_t<float>(dtype, n, ops, count, mode, segments, source_bytes, code_bytes, &first));
  if (first_source) *first_source = first.c_str();
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_tile_bits(void) { return kTileBits; }

// option "tile_auto": from this size a program compiles its segments (below it a sweep is launch-bound and the interpreter, replayed
// as a graph, is as fast: profiles/r01_small_n_launch_bound.md)
int64_t g_debug_slice_sweeps = 0;  // global option "debug_slice_sweeps" (measuring aid, see apply_ops_tiled)
constexpr int kAutoJitMinQubits = 22;
// ... and from this size a one-shot apply_ops looks its wide plan up (r6).  The lookup plans twice and fingerprints every segment:
// ~3.5 us of host time per gate, which at n = 22 makes the compiled path host-bound and SLOWER than the interpreter (2000 gates:
// 7.4 against 6.2 ms); from n = 24 it wins (10.0 against 15.4 ms; n = 26: 36 against 57 ms).
constexpr int kAutoOneShotMinQubits = 24;
constexpr int kPairFloorMinQubits = 22;  // option "pair_floor": only where a sweep is HBM-bound (below, launches are)
static bool tile_wide_of(const qip_hip_state* s) {  // wide tiles: run-time-compiled segments only, a state above one wide tile
  // (r5: also inside a graph capture — the launch and its parameter upload are ordinary stream work like the 11-bit segments')
  return s->tile_wide && s->tile_jit && s->tile_passes && s->n > (uint32_t)kWideBits;
}
static int tile_mode_of(const qip_hip_state* s) {  // option tile_relabel: 1 = when it shortens the plan, 2 = always
  return (int)std::min<int64_t>(s->tile, 2) | (s->tile_relabel ? 4 : 0) | (s->tile_relabel == 2 ? 8 : 0) |  // (3 = persistent layout)
         (tile_wide_of(s) ? 16 : 0);
}

int state_tile_mode(const qip_hip_state* s) { return tile_mode_of(s); }  // (qip_dist.hip schedules the edge sweeps of a remap with it)

template <typename T>
static int apply_ops_tiled(qip_hip_state* s, const qip_op* ops_in, uint64_t count, bool /*reorder*/) {
  TileSchedule sc;
  // tile_relabel = 3: the qubit layout persists across calls — the plan starts from the layout the previous call left and
  // does not close with the restoring sweep (that happens once, when somebody needs the caller's order: state_settle).
  // Not inside a graph capture or a compile-only pass: a recorded program must start and end in the caller's order.
  const bool persist = s->tile_relabel >= 3 && !s->capture_pool && !s->jit_prepare;
  if (!persist && !s->layout.empty()) QCHK(state_settle(s));
  sc.init_phys = s->layout;
  sc.keep_layout = persist;
  QCHK(make_tile_schedule(s->dtype, s->n, ops_in, count, tile_mode_of(s), s->tile_passes != 0, &sc));
  {
    // a bit-permutation sweep is out of place: get the second buffer BEFORE the first gate runs; if HBM cannot hold it
    // (a state above half of the 288 GB), fall back to the plan without permutation sweeps instead of failing half way
    bool permutes = false;
    for (const TileStep& st : sc.steps) permutes = permutes || !st.perm.empty();
    // ... and a plan that LEAVES the state relabelled (tile_relabel = 3) commits the first reader to a settling sweep, which
    // needs that buffer too: a state too large for it would become unreadable — same fallback
    bool leaves_relabelled = false;
    if (persist)
      for (uint32_t p = 0; p < sc.final_phys.size(); ++p) leaves_relabelled = leaves_relabelled || sc.final_phys[p] != p;
    if ((permutes || leaves_relabelled) && !s->jit_prepare && !s->alt && ensure_alt(s) != QIP_OK) {
      (void)hipGetLastError();
      if (!s->layout.empty()) return fail(QIP_ERR_DEVICE, "no room for the second buffer a relabelled state needs");
      sc = TileSchedule();
      // (without relabelling: such a plan neither permutes nor leaves a layout behind)
      QCHK(make_tile_schedule(s->dtype, s->n, ops_in, count, tile_mode_of(s) & (3 | 16), s->tile_passes != 0, &sc, /*allow_permute=*/false));
    }
  }
  const qip_op* ops = sc.circuit;
  const std::vector<TileItem>& items = sc.items;
  // from here on the shard is somewhere between two layouts until the last step has been issued: single ops inside the plan
  // are already expressed in physical positions, so they must not settle (layout is cleared for the duration)
  const std::vector<uint32_t> final_phys = sc.final_phys;
  // a relabelled plan moves the qubits between its steps: if a step fails (a launch, the arena, the run-time compiler) the
  // buffer is in an order no caller can name — the handle is poisoned until it is re-initialised (the sharded
  // handle does the same).  A plain plan that fails leaves a prefix of the circuit applied in the caller's order, like the
  // gate-by-gate path: reported, not poisoned.
  bool moves_qubits = !sc.init_phys.empty() || sc.inserted > 0 || sc.absorbed > 0;
  for (const TileStep& st : sc.steps) moves_qubits = moves_qubits || !st.perm.empty();
  const bool wide = (tile_mode_of(s) & 16) != 0;
  // (the pre-compilation runs BEFORE the layout is cleared — a compiler failure here returns with the handle
  // exactly as it was, still naming the layout the data is in)
  // r5: every segment of the plan that is not resident yet is found FIRST (a compile-only pass that collects the sources), looked
  // up in the disk cache, and the rest compiled side by side in helper processes (jit_compile_collected) before anything runs.
  // A plan that has been through this once is remembered by a fingerprint of its structure, so the steady state (the same
  // circuit applied again and again) does not generate every source twice.
  if (s->tile_jit && s->tile_passes && !s->capture_pool) {
    uint64_t fp = 1469598103934665603ull;
    auto mix = [&](uint64_t v) {
      fp ^= v;
      fp *= 1099511628211ull;
    };
    mix(s->n); mix((uint64_t)s->dtype); mix((uint64_t)s->device); mix((uint64_t)tile_mode_of(s)); mix((uint64_t)s->tile_jit);
    mix((uint64_t)(s->tile_fma != 0) | (uint64_t)(s->tile_merge != 0) << 1 | (uint64_t)(g_tile_wide_pin != 0) << 2 |
        (uint64_t)(g_tile_wide_dense3_inline != 0) << 3 | (uint64_t)use_nt(s) << 4 | (uint64_t)g_tile_remap << 8);
    for (const TileStep& st : sc.steps) {
      if (st.ops.size() < 2 || !st.perm.empty()) continue;
      mix(0xABCDu + st.ops.size());
      for (uint32_t h : st.high) mix(h);
      for (uint64_t i : st.ops) {
        const TileItem& it = items[i];
        mix((uint64_t)it.kind | (uint64_t)it.nz << 8 | (uint64_t)it.exact << 40);
        mix(it.nd_mask);
        mix(it.d_mask);
        for (uint32_t c : it.cpos) mix(0x100u + c);
        mix(it.t0 | (uint64_t)it.t1 << 8 | (uint64_t)it.t2 << 16);
      }
    }
    bool warm;
    {
      std::lock_guard<std::mutex> lock(g_jit_mutex);
      warm = g_jit_warm_plans.count(fp) != 0 && g_jit_warm_generation == g_jit_generation;
      if (!warm && s->jit_lookup_only) {  // a plan seen incomplete a moment ago: its segments are still being compiled, do not look again yet
        auto it = g_jit_cold_plans.find(fp);
        if (it != g_jit_cold_plans.end() && std::chrono::steady_clock::now() - it->second < std::chrono::milliseconds(1500)) return kJitMiss;
      }
    }
    if (!warm) {
      std::vector<std::pair<std::string, bool>> jobs;
      const bool outer_prepare = s->jit_prepare;
      s->jit_collect = &jobs;
      s->jit_prepare = true;
      int rc = QIP_OK;
      for (const TileStep& st : sc.steps) {
        if (st.ops.size() < 2 || !st.perm.empty()) continue;
        std::vector<const TileItem*> seg;
        for (uint64_t i : st.ops) seg.push_back(&items[i]);
        rc = wide ? launch_wide_segment<T>(s, seg, st.high) : launch_tile_segment<T>(s, seg, st.high);
        if (rc != QIP_OK) break;
      }
      s->jit_prepare = outer_prepare;
      s->jit_collect = nullptr;
      QCHK(rc);
      if (s->jit_lookup_only) {
        std::vector<std::pair<std::string, bool>> misses;
        QCHK(jit_load_only(s, jobs, &misses));
        if (!misses.empty()) {
          jit_compile_in_background(misses);
          std::lock_guard<std::mutex> lock(g_jit_mutex);
          if (g_jit_cold_plans.size() > 1024) g_jit_cold_plans.clear();
          g_jit_cold_plans[fp] = std::chrono::steady_clock::now();
          return kJitMiss;  // (nothing has run, the layout is untouched)
        }
      } else {
        QCHK(jit_compile_collected(s, jobs));
      }
      std::lock_guard<std::mutex> lock(g_jit_mutex);
      if (g_jit_warm_generation != g_jit_generation || g_jit_warm_plans.size() > 4096) {  // (an eviction may have dropped what a warm plan needs)
        g_jit_warm_plans.clear();
        g_jit_warm_generation = g_jit_generation;
      }
      g_jit_warm_plans.insert(fp);
    }
  }
  s->layout.clear();
  // The multi-GPU gather may ride in the last sweep's store phase only when that sweep addresses the CALLER's bit order: the
  // request names caller-order positions (TileStorePerm.sel).  A plan that leaves the qubits relabelled (tile_relabel = 3 with
  // a non-identity final layout) ends in physical positions that mean other qubits: no fold, the layout is settled and the
  // gather runs as a sweep of its own (qip_dist.hip dist_run_steps).
  bool ends_in_callers_order = true;
  for (uint32_t p = 0; p < final_phys.size(); ++p) ends_in_callers_order = ends_in_callers_order && final_phys[p] == p;
  auto run_steps = [&]() -> int {
    size_t step_no = 0;
    // a first-step request that no step can take (an empty plan; one step that already serves the last-step request) is settled
    // BEFORE anything is enqueued: its fallback is what makes the data the batch reads complete
    if (s->slice_first && !s->jit_prepare && !s->capture_pool && (sc.steps.empty() || (sc.steps.size() == 1 && s->slice_last && ends_in_callers_order))) {
      TileSlicing* w = s->slice_first;
      s->slice_first = nullptr;
      if (w->fallback) QCHK(w->fallback());
    }
    for (const TileStep& st : sc.steps) {
      ++step_no;
      s->fold_now = s->fold_request && ends_in_callers_order && step_no == sc.steps.size();  // (the batch's last step may store packed)
      // r5: the first / last step in parts (the sharded state's overlapped exchange); positions are caller-order positions, so
      // only a plan that starts / ends in the caller's order may take the request.  A step that is not a multi-gate tile sweep
      // (or a sweep that cannot be cut there) consumes it through its fallback.
      TileSlicing* want = nullptr;
      if (!s->jit_prepare && !s->capture_pool) {
        if (step_no == sc.steps.size() && s->slice_last && ends_in_callers_order) want = s->slice_last;
        else if (step_no == 1 && s->slice_first && sc.init_phys.empty() && sc.inserted == 0 && sc.absorbed == 0) want = s->slice_first;
        else if (step_no == 1 && s->slice_first) {  // (a relabelled plan addresses other positions: settle the request unsliced)
          TileSlicing* w = s->slice_first;
          s->slice_first = nullptr;
          if (w->fallback) QCHK(w->fallback());
        }
      }
      TileSlicing dbg;
      if (!want && g_debug_slice_sweeps >= 2 && !s->jit_prepare && !s->capture_pool && st.perm.empty() && st.ops.size() >= 2) {
        // measuring aid (global option "debug_slice_sweeps" = 2 / 4 / 8): EVERY multi-gate sweep in that many parts, cut at the highest
        // index positions its tile leaves alone — what a sliced launch costs by itself, at full size on one GPU, without any exchange
        uint32_t pb = 0;
        while ((1 << pb) < g_debug_slice_sweeps) ++pb;
        dbg.nbits = 0;
        for (uint32_t pp = s->n; pp-- > 12u && dbg.nbits < pb;)
          if (std::find(st.high.begin(), st.high.end(), pp) == st.high.end()) dbg.pos[dbg.nbits++] = pp;
        if (dbg.nbits == pb) want = &dbg;
      }
      s->slice_now = want;
      auto unsliced = [&]() -> int {
        TileSlicing* w = s->slice_now;
        s->slice_now = nullptr;
        return (w && w->fallback) ? w->fallback() : QIP_OK;
      };
      if (!st.perm.empty()) {  // a run of Swap ops as one bit-permutation sweep
        if (s->jit_prepare) continue;
        QCHK(unsliced());
        QCHK(launch_permute(s, st.perm.data()));
        continue;
      }
      if (st.ops.size() == 1) {
        QCHK(unsliced());
        QCHK(apply_op_t<T>(s, &ops[st.ops[0]]));
        continue;
      }
      std::vector<const TileItem*> seg;
      for (uint64_t i : st.ops) seg.push_back(&items[i]);
      if (wide) QCHK(launch_wide_segment<T>(s, seg, st.high));
      else QCHK(launch_tile_segment<T>(s, seg, st.high));
    }
    return QIP_OK;
  };
  const int rc_steps = run_steps();
  s->fold_now = false;
  s->slice_now = nullptr;
  if (rc_steps != QIP_OK) {
    if (moves_qubits && !s->jit_prepare) {
      s->poisoned = true;
      s->poison_msg = g_last_error;
    }
    return rc_steps;
  }
  if (persist) {
    bool identity = true;
    for (uint32_t p = 0; p < final_phys.size(); ++p) identity = identity && final_phys[p] == p;
    if (!identity) s->layout = final_phys;
  }
  return QIP_OK;
}

extern "C" int qip_hip_state_apply_ops(qip_hip_state* s, const qip_op* ops, uint64_t count) try {
  STATE_ENTER_RAW(s);  // (a relabelled state stays relabelled for a relabelling batch: apply_ops_tiled decides)
  if (count && !ops) return fail(QIP_ERR_INVALID, "null op array");
  if (s->tile >= 1 && !s->force_generic && !g_force_generic && s->n >= (uint32_t)kTileBits) {
    // r6, option "tile_auto" for one-shot callers: the compiled (wide) sweeps when every segment of the plan is already resident
    // or in the disk cache; otherwise the interpreter now, the misses compiled in the background for the next call.  Not for the
    // batches of a sharded state that carry a fold / slice request (their last sweep is chosen by the exchange), not while a
    // program records or pre-compiles (it compiles for itself).
    if (s->tile_auto && !s->tile_jit && s->tile_passes && s->n >= (uint32_t)kAutoOneShotMinQubits && count >= 2 && !s->capture_pool && !s->jit_prepare &&
        !s->fold_request && !s->slice_first && !s->slice_last && !s->profile && g_jit_disk) {
      const int64_t wide0 = s->tile_wide;
      s->tile_jit = 1;
      s->tile_wide = s->n > (uint32_t)kWideBits ? 1 : 0;
      s->jit_lookup_only = true;
      const int rc = s->dtype == QIP_C64 ? apply_ops_tiled<double>(s, ops, count, s->tile >= 2) : apply_ops_tiled<float>(s, ops, count, s->tile >= 2);
      s->jit_lookup_only = false;
      s->tile_jit = 0;
      s->tile_wide = wide0;
      if (rc != kJitMiss) return rc;
    }
    return s->dtype == QIP_C64 ? apply_ops_tiled<double>(s, ops, count, s->tile >= 2)
                               : apply_ops_tiled<float>(s, ops, count, s->tile >= 2);
  }
  if (s->slice_first) {  // (only tile sweeps run in parts: every other path settles the request before its first launch)
    TileSlicing* w = s->slice_first;
    s->slice_first = nullptr;
    if (w->fallback) QCHK(w->fallback());
  }
  if (!s->layout.empty()) QCHK(state_settle(s));
  if (s->fuse >= 2 && !s->force_generic && !g_force_generic) {
    const uint32_t K = (uint32_t)std::min<int64_t>(s->fuse, kMaxMfmaK);  // both precisions have a matrix-core k = 5 kernel
    if (s->n >= K + 4)
      return s->dtype == QIP_C64 ? apply_ops_fused<double>(s, ops, count, K) : apply_ops_fused<float>(s, ops, count, K);
  }
  for (uint64_t i = 0; i < count; ++i) {
    // r5, option "pair_floor" (default 1): a gate whose selectors (controls, the target of a phase-type diagonal) sit inside a
    // 1-KiB wave row sweeps the WHOLE vector for half / a quarter of the algorithmic bytes — the memory system moves whole lines
    // (T on bit 0: 41 %, CNOT with its control inside a row: 40 %, profiles/r02_line_bits.md).  When such a gate and its
    // neighbour fit one tile they go as ONE two-item tile sweep (interpreter kernel, circuit order: the same unfused arithmetic
    // per amplitude, IEEE-equal to the two launches) — the neighbour rides for free.  Everything else stays one launch per gate.
    if (s->pair_floor && i + 1 < count && s->n >= (uint32_t)kPairFloorMinQubits && s->tile_passes && !s->capture_pool && !s->force_generic &&
        !g_force_generic) {
      // What the two launches move, in sweeps of the whole vector: a gate's algorithmic share, doubled for every selector
      // inside a wave row (whole lines / rows travel whichever half is needed), at most 1.  One two-item sweep moves 1 (and runs
      // a little slower than a bare sweep): worth it from 1.3 — T on a low bit + H (1 + 1), CNOT with a low control + anything;
      // NOT two controlled phases of a QFT (1/4 doubled = 1/2 each: measured 946 -> 986 ms when they were paired).
      bool floor_gate = false, both = true;
      double moved = 0;
      const double full = 2.0 * (double)s->amp_bytes * (double)s->namps;
      for (int j = 0; j < 2 && both; ++j) {
        TileItem it;
        if (classify_tile_item(s->dtype, s->n, &ops[i + j], &it) != QIP_OK || !it.tileable) {
          both = false;
          break;
        }
        double by = 0;
        if (qip_hip_op_algorithmic_bytes(s->dtype, s->n, &ops[i + j], &by) != QIP_OK) {
          both = false;
          break;
        }
        const int in_row = __builtin_popcountll(it.d_mask & 63ull);
        const double phys = std::min(1.0, by / full * (double)(1u << in_row));
        if (in_row && phys > by / full) floor_gate = true;
        moved += phys;
      }
      if (both && floor_gate && moved >= 1.3) {
        TileSchedule sc;
        if (make_tile_schedule(s->dtype, s->n, &ops[i], 2, 1, true, &sc) == QIP_OK && sc.steps.size() == 1 && sc.steps[0].ops.size() == 2 &&
            sc.steps[0].perm.empty()) {
          const int64_t tile = s->tile, jit = s->tile_jit, relabel = s->tile_relabel, wide = s->tile_wide;
          s->tile = 1;
          s->tile_jit = s->tile_relabel = s->tile_wide = 0;
          // the sharded state's requests belong to the BATCH's last step, not to this pair's: a pair in the middle of the batch
          // must not store packed (the gates after it would run on the packed buffer) — only the batch's last two ops may
          const TileStorePerm* fold_req = s->fold_request;
          TileSlicing *sf = s->slice_first, *sl = s->slice_last;
          if (i + 2 < count) s->fold_request = nullptr;
          s->slice_first = s->slice_last = nullptr;
          const int rc = s->dtype == QIP_C64 ? apply_ops_tiled<double>(s, &ops[i], 2, false) : apply_ops_tiled<float>(s, &ops[i], 2, false);
          s->fold_request = fold_req;
          s->slice_first = sf;
          s->slice_last = sl;
          s->tile = tile;
          s->tile_jit = jit;
          s->tile_relabel = relabel;
          s->tile_wide = wide;
          if (rc != QIP_OK) {
            std::string msg = g_last_error;
            return fail(rc, "ops %llu, %llu: %s", (unsigned long long)i, (unsigned long long)i + 1, msg.c_str());
          }
          ++i;
          continue;
        }
        (void)hipGetLastError();
      }
    }
    s->fold_now = s->fold_request && i + 1 == count;  // (a last op that runs as a one-op tile sweep may store packed)
    int rc = s->dtype == QIP_C64 ? apply_op_t<double>(s, &ops[i]) : apply_op_t<float>(s, &ops[i]);
    s->fold_now = false;
    if (rc != QIP_OK) {
      std::string msg = g_last_error;
      return fail(rc, "op %llu: %s", (unsigned long long)i, msg.c_str());
    }
  }
  return QIP_OK;
} QIP_CATCH_ALL

// ---------------------------------------------------------------------------------------
// programs: a circuit PREPARED once — every op's tables packed and resident in device memory the program owns — and
// recorded into a hipGraph: a run is one graph launch of kernel nodes only (nothing is packed, uploaded or decided again).
// Ops that write the second buffer (the literal gather, k_sparse_ell, bit-permutation sweeps) are recorded like the rest:
// the ping-pong is deterministic, so a program keeps one recording per starting buffer (at most two) and follows the
// buffers on the host after every launch, the way the reference swaps `state` and `arena` (builder.rs:514).
// ---------------------------------------------------------------------------------------
struct ProgRecording {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  void *start_cur = nullptr, *start_alt = nullptr;  // the buffers the recording reads first / writes first
  bool swaps = false;                                // an odd number of out-of-place launches: the buffers end exchanged
  bool uses_alt = false;                             // some launch writes the second buffer
  ProgPool pool;                                     // the device copy of the ops' payloads
  uint64_t jit_gen = 0;                              // run-time-compiled kernels the cache may since have unloaded
};
struct qip_hip_program {
  qip_hip_state* s = nullptr;
  const qip_op* ops = nullptr;
  uint64_t count = 0;
  ProgRecording rec[2];
  bool capture_failed = false;  // an obstacle that will not go away (profile / force_generic aside): stay eager, do not retry every run
  int last_was_graph = 0;
  // r5, option "tile_auto": a program is made to be replayed, so it repays a compilation — created on a state with tile >= 1 and
  // tile_jit = 0 (the interpreter, what apply_ops keeps using) it runs its own launches with run-time-compiled wide segments.
  // Same arithmetic, same order: the
  // results are bit-identical to the interpreter's (tile = 1) / within the mode's own bar (tile = 2).
  bool auto_jit = false, auto_wide = false;
};
// the program's own options, in force while one of ITS launches (capture, replay, eager run) is issued
struct ProgramOptions {
  qip_hip_state* s;
  int64_t jit, wide;
  bool on;
  explicit ProgramOptions(qip_hip_program* p) : s(p->s), jit(0), wide(0), on(p->s && p->auto_jit) {
    if (!on) return;
    jit = s->tile_jit;
    wide = s->tile_wide;
    s->tile_jit = 1;
    s->tile_wide = p->auto_wide ? 1 : 0;
  }
  ~ProgramOptions() {
    if (!on) return;
    s->tile_jit = jit;
    s->tile_wide = wide;
  }
};

static void recording_drop(ProgRecording* r) {
  if (r->exec) (void)hipGraphExecDestroy(r->exec);
  if (r->graph) (void)hipGraphDestroy(r->graph);
  if (r->pool.base) (void)hipFree(r->pool.base);
  *r = ProgRecording();
}
static void program_drop_graph(qip_hip_program* p) {
  for (ProgRecording& r : p->rec) recording_drop(&r);
}
void programs_orphan(qip_hip_state* s) {
  for (qip_hip_program* p : s->programs) {
    program_drop_graph(p);
    p->s = nullptr;
  }
  s->programs.clear();
}

// Record the program for the state's CURRENT buffers into `r`; on any obstacle leave `r` empty (the run is eager then) and report
// success — only a real descriptor error is returned.
static int program_capture(qip_hip_program* p, ProgRecording* r) {
  qip_hip_state* s = p->s;
  ProgramOptions scope(p);
  recording_drop(r);
  if (s->force_generic || g_force_generic || s->profile) return QIP_OK;
  if (!s->stream) return QIP_OK;  // (the legacy default stream of a wrapped state cannot be captured)
  // the second buffer must exist BEFORE the recording starts (no allocation inside a stream capture): ask for it when any launch
  // of the plan writes out of place; a state too large for it stays eager (and fails there with the allocation's own message)
  bool out_of_place = false;
  for (uint64_t i = 0; i < p->count && !out_of_place; ++i) {
    FlatOp f;
    QCHK(flatten_op(s->n, &p->ops[i], false, &f));
    Plan pl;
    QCHK(make_plan(s->dtype, s->n, f, false, &pl));
    const uint32_t k = f.n_op;
    const bool reg_or_mfma = pl.cls == KC_GATE_KQ && (k <= kMaxRegK || (s->mfma && k <= kMaxHugeK && s->n >= f.k_all + 4));
    if (pl.cls == KC_GATHER_GENERIC && sparse_tile_applies(s, pl, f)) continue;  // (in place through k_sparse_tile)
    out_of_place = pl.cls == KC_GATHER_GENERIC || (pl.cls == KC_GATE_KQ && !reg_or_mfma);
  }
  if (!out_of_place && s->tile >= 1 && s->n >= (uint32_t)kTileBits) {  // a bit-permutation sweep is out of place too
    TileSchedule sc;
    QCHK(make_tile_schedule(s->dtype, s->n, p->ops, p->count, tile_mode_of(s), s->tile_passes != 0, &sc));
    for (const TileStep& st : sc.steps) out_of_place = out_of_place || !st.perm.empty();
  }
  if (out_of_place && !s->alt && ensure_alt(s) != QIP_OK) {
    (void)hipGetLastError();
    p->capture_failed = true;
    return QIP_OK;
  }
  if (s->tile >= 1 && s->tile_jit) {  // run-time compilation cannot happen inside a stream capture: do it now
    s->jit_prepare = s->jit_for_capture = true;
    const int rc = qip_hip_state_apply_ops(s, p->ops, p->count);
    s->jit_prepare = s->jit_for_capture = false;
    QCHK(rc);
  }
  void* const arena0 = s->arena;
  const size_t arena_cap0 = s->arena_cap;
  void *const cur0 = s->cur, *const alt0 = s->alt;
  const bool owns_cur0 = s->owns_cur, owns_alt0 = s->owns_alt;
  ProgPool& pool = r->pool;
  int result = QIP_OK;
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (!pool.base) {
      pool.cap = std::max<size_t>(pool.cap, 1 << 16);
      if (hipMalloc(&pool.base, pool.cap) != hipSuccess) {
        (void)hipGetLastError();
        pool = ProgPool();
        p->capture_failed = true;
        break;
      }
    }
    pool.used = 0;
    pool.overflow = false;
    pool.image.clear();
    if (hipStreamBeginCapture(s->stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
      (void)hipGetLastError();
      p->capture_failed = true;
      break;
    }
    s->capture_pool = &pool;
    jit_capture_scope(+1);
    const int rc = qip_hip_state_apply_ops(s, p->ops, p->count);
    s->capture_pool = nullptr;
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(s->stream, &g);
    jit_capture_scope(-1);
    // the recording ISSUED nothing: the host's view of the buffers goes back to where it was (the launch will move it)
    const bool swapped = s->cur != cur0;
    s->cur = cur0;
    s->alt = alt0;
    s->owns_cur = owns_cur0;
    s->owns_alt = owns_alt0;
    s->arena = arena0;
    s->arena_cap = arena_cap0;
    if (rc == QIP_OK && e == hipSuccess && g && !pool.overflow) {
      bool ok = true;
      if (pool.used) {
        pool.image.resize(pool.used);
        ok = hipMemcpy(pool.base, pool.image.data(), pool.used, hipMemcpyHostToDevice) == hipSuccess;
      }
      pool.image.clear();
      pool.image.shrink_to_fit();
      if (ok && hipGraphInstantiate(&r->exec, g, nullptr, nullptr, 0) == hipSuccess) {
        r->graph = g;
        r->start_cur = cur0;
        r->start_alt = alt0;
        r->swaps = swapped;
        r->uses_alt = out_of_place;
        r->jit_gen = jit_cache_generation();
        return QIP_OK;
      }
      (void)hipGetLastError();
      (void)hipGraphDestroy(g);
      r->exec = nullptr;
      p->capture_failed = true;
      break;
    }
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    if (rc == QIP_OK && e == hipSuccess && pool.overflow) {  // the sizing pass: now the pool's size is known — once more
      (void)hipFree(pool.base);
      pool.base = nullptr;
      pool.cap = pool.used + 4096;
      continue;
    }
    if (rc != QIP_OK && rc != QIP_ERR_UNSUPPORTED) result = rc;  // a real descriptor error
    else p->capture_failed = true;
    break;
  }
  recording_drop(r);
  return result;
}

extern "C" int qip_hip_program_create(qip_hip_state* s, const qip_op* ops, uint64_t count,
                                      qip_hip_program** out) try {
  STATE_ENTER(s);
  if (!out || (count && !ops)) return fail(QIP_ERR_INVALID, "null argument");
  for (uint64_t i = 0; i < count; ++i) {
    FlatOp f;
    int rc = flatten_op(s->n, &ops[i], false, &f);
    if (rc != QIP_OK) {
      std::string msg = g_last_error;
      return fail(rc, "op %llu: %s", (unsigned long long)i, msg.c_str());
    }
  }
  qip_hip_program* p = new qip_hip_program();
  p->s = s;
  p->ops = ops;
  p->count = count;
  if (s->tile_auto && s->tile >= 1 && !s->tile_jit && s->tile_passes && s->n >= (uint32_t)kAutoJitMinQubits && !s->force_generic && !g_force_generic) {
    p->auto_jit = true;
    p->auto_wide = s->n > (uint32_t)kWideBits;  // (dense 3-qubit items included since they are written out group by group: tile_wide_dense3_inline)
  }
  int rc = program_capture(p, &p->rec[0]);
  if (rc != QIP_OK && p->auto_jit) {
    // the automatic choice must not turn a compiler problem (libhiprtc missing, a hiprtc error on one segment) into a
    // failed program_create — without it the interpreter would have served.  Drop the automatic options and record again with the
    // state's own; only options the CALLER set propagate their errors.
    p->auto_jit = p->auto_wide = false;
    p->capture_failed = false;
    rc = program_capture(p, &p->rec[0]);
  }
  if (rc != QIP_OK) {
    program_drop_graph(p);
    delete p;
    return rc;
  }
  s->programs.push_back(p);
  *out = p;
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_program_run(qip_hip_program* p) try {
  if (!p) return fail(QIP_ERR_INVALID, "null program");
  qip_hip_state* s = p->s;
  if (!s) return fail(QIP_ERR_INVALID, "the state this program was recorded against has been destroyed");
  STATE_ENTER(s);
  ProgramOptions scope(p);
  if (s->profile || s->force_generic || g_force_generic) {  // (profiling brackets every kernel with events; the literal kernel is a test route)
    p->last_was_graph = 0;
    return qip_hip_state_apply_ops(s, p->ops, p->count);
  }
  for (int attempt = 0; attempt < 2; ++attempt) {
    // the recording made for the buffers as they are now; none yet (the previous run left the state on its other buffer, an eager
    // op exchanged them): record into the free slot, or over the one that does not start here
    ProgRecording* r = nullptr;
    for (ProgRecording& c : p->rec)
      if (c.exec && c.start_cur == s->cur && (!c.uses_alt || c.start_alt == s->alt)) r = &c;
    if (!r) {
      if (p->capture_failed) break;
      ProgRecording* slot = !p->rec[0].exec ? &p->rec[0] : (!p->rec[1].exec ? &p->rec[1] : (p->rec[0].start_cur == s->alt ? &p->rec[1] : &p->rec[0]));
      QCHK(program_capture(p, slot));
      if (!slot->exec) break;
      r = slot;
    }
    // generation check and launch in ONE critical section: an eviction on another thread between the two would unload a
    // module this graph names.  Enqueueing is asynchronous: the section is microseconds.
    bool stale = false;
    {
      std::lock_guard<std::mutex> lock(g_jit_mutex);
      stale = s->tile_jit && r->jit_gen != g_jit_generation;
      if (!stale) HIPCHK(hipGraphLaunch(r->exec, s->stream));
    }
    if (stale) {
      recording_drop(r);
      continue;
    }
    if (r->swaps) {  // builder.rs:514
      std::swap(s->cur, s->alt);
      std::swap(s->owns_cur, s->owns_alt);
    }
    p->last_was_graph = 1;
    return QIP_OK;
  }
  p->last_was_graph = 0;
  return qip_hip_state_apply_ops(s, p->ops, p->count);
} QIP_CATCH_ALL

extern "C" int qip_hip_program_is_graph(const qip_hip_program* p) try { return p ? p->last_was_graph : 0; } QIP_CATCH_ALL

extern "C" int qip_hip_program_destroy(qip_hip_program* p) try {
  if (!p) return QIP_OK;
  if (p->s) {
    (void)hipSetDevice(p->s->device);
    (void)hipStreamSynchronize(p->s->stream);
    auto& v = p->s->programs;
    v.erase(std::remove(v.begin(), v.end(), p), v.end());
  }
  program_drop_graph(p);
  delete p;
  return QIP_OK;
} QIP_CATCH_ALL
