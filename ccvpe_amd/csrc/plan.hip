// ccvpe_ctx / ccvpe_forward: the WHOLE eval forward behind one C entry point ("opaque ccvpe_ctx per
// (model kind, B, grd H x W, N_rot set, dtype) created once; ccvpe_forward(ctx, grd, sat, out[9], stream)").
//
// A ctx is built from a PLAN: the launch list of one forward (models.py:150-343 / :448-652 / :752-950 as ccvpe_amd/models.py
// runs it) — every C-ABI call of this library with its arguments, where each pointer is an offset into one of four regions:
//   WEIGHTS    the packed weights (BN folded, K-major panels, folded deconv + conv, ...), one blob, part of the plan;
//   WORKSPACE  every intermediate and the nine outputs, laid out by lifetime (a buffer's bytes are reused once it is dead);
//   GRD / SAT  the caller's input images (NCHW fp32), given per call.
// ccvpe_amd/plan.py records the plan by running the Python forward once against a recording allocator and serialises it; the
// bytes can be written to a file, so a caller WITHOUT Python (tools/plan_run.cpp) loads them, creates a ctx and runs.
// The forward itself is a loop over the recorded calls: no Python, no per-call descriptor building, no allocation.  Calls carry the
// stream they were recorded on — the caller's stream, or the ctx's own side stream (ground encoder beside the aerial encoder,
// and in bf16 storage the orientation decoder beside the localisation decoder) — with the recorded fork / join waits replayed
// through events, so the whole forward is ordered after and before the caller's stream (hipGraph-capturable).  Host-only code: the kernels are the library's own entry points.
#include <cstring>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "common.h"

namespace ccvpe {

enum : uint32_t { K_INT = 0, K_FLT = 1, K_NULL = 2, K_WEIGHTS = 3, K_WORKSPACE = 4, K_GRD = 5, K_SAT = 6, K_STREAM = 7, K_BLOB = 8 };

template <typename A> struct FromSlot {                      // integers
  static A get(uint64_t v) { return (A)(int64_t)v; }
};
template <typename P> struct FromSlot<P*> {
  static P* get(uint64_t v) { return reinterpret_cast<P*>((uintptr_t)v); }
};
template <> struct FromSlot<float> {
  static float get(uint64_t v) { uint32_t b = (uint32_t)v; float f; std::memcpy(&f, &b, 4); return f; }
};
template <typename... A, size_t... I>
static int invoke_seq(int (*f)(A...), const uint64_t* v, std::index_sequence<I...>) { return f(FromSlot<A>::get(v[I])...); }
template <typename... A>
static int invoke(int (*f)(A...), const uint64_t* v, int n) {
  if (n != (int)sizeof...(A)) return fail(CCVPE_EINVAL, "plan: call has %d arguments, the entry point takes %d", n, (int)sizeof...(A));
  return invoke_seq(f, v, std::index_sequence_for<A...>{});
}
typedef int (*Invoker)(const uint64_t*, int);

// every entry point an eval forward can contain
#define CCVPE_REG(fn) {#fn, [](const uint64_t* v, int n) -> int { return invoke(fn, v, n); }}
static const std::unordered_map<std::string, Invoker>& registry() {
  static const std::unordered_map<std::string, Invoker> r = {
      CCVPE_REG(ccvpe_conv_igemm_f32), CCVPE_REG(ccvpe_conv_igemm_bf16), CCVPE_REG(ccvpe_conv_igemm_splitk_f32),
      CCVPE_REG(ccvpe_conv_igemm_splitk_bf16), CCVPE_REG(ccvpe_conv3x3_match1_bf16), CCVPE_REG(ccvpe_upconv3x3_f32), CCVPE_REG(ccvpe_upconv3x3_bf16),
      CCVPE_REG(ccvpe_tail512_f32), CCVPE_REG(ccvpe_tail512_bf16), CCVPE_REG(ccvpe_stem_conv_f32), CCVPE_REG(ccvpe_stem_conv_bf16),
      CCVPE_REG(ccvpe_stem_dw_f32), CCVPE_REG(ccvpe_stem_dw_bf16),
      CCVPE_REG(ccvpe_dwconv_f32), CCVPE_REG(ccvpe_dwconv_bf16), CCVPE_REG(ccvpe_mbconv_front_f32), CCVPE_REG(ccvpe_mbconv_front_bf16),
      CCVPE_REG(ccvpe_se_gate_f32), CCVPE_REG(ccvpe_ground_descriptor_f32), CCVPE_REG(ccvpe_match_level_f32),
      CCVPE_REG(ccvpe_match_level_bf16), CCVPE_REG(ccvpe_head_conv3x3_f32), CCVPE_REG(ccvpe_head_conv3x3_bf16),
      CCVPE_REG(ccvpe_softmax_rows_f32), CCVPE_REG(ccvpe_softmax_apply_f32), CCVPE_REG(ccvpe_cast_bf16_f32), CCVPE_REG(ccvpe_eval_postprocess_f32),
  };
  return r;
}
#undef CCVPE_REG

struct Patch { uint32_t call, arg, kind; uint64_t off; };              // per-run argument (input image / stream)
struct BlobPatch { uint32_t blob, field, kind; uint64_t off; };        // per-run pointer field inside a descriptor blob
struct Call { Invoker fn; std::string name; uint32_t first, nargs, sid; int wait_on; };   // fn == nullptr: "@wait" (stream sid waits for stream wait_on)
struct Output { uint64_t off, bytes; uint32_t dtype, ndim; long long dims[4], strides[4]; };

}  // namespace ccvpe

struct ccvpe_ctx {
  std::vector<ccvpe::Call> calls;
  std::vector<uint64_t> args;                       // resolved argument values of all calls, back to back
  std::vector<std::vector<unsigned char>> blobs;    // descriptor structs / host arrays the calls point to
  std::vector<ccvpe::Patch> patches;
  std::vector<ccvpe::BlobPatch> blob_patches;
  std::vector<ccvpe::Output> outputs;
  uint64_t workspace_bytes = 0, weights_bytes = 0, grd_bytes = 0, sat_bytes = 0;
  char* weights = nullptr;
  char* workspace = nullptr;
  bool own_weights = false, own_workspace = false;
  hipStream_t side = nullptr;                       // the forward's second stream (ground encoder / orientation decoder)
  std::vector<hipEvent_t> events;                   // one per recorded stream wait
};

using namespace ccvpe;

namespace {
struct Reader {
  const unsigned char* p;
  const unsigned char* end;
  bool ok = true;
  template <typename T> T get() {
    T v{};
    if (sizeof(T) > (size_t)(end - p)) { ok = false; return v; }
    std::memcpy(&v, p, sizeof(T));
    p += sizeof(T);
    return v;
  }
  const unsigned char* bytes(size_t n) {
    if (n > (size_t)(end - p)) { ok = false; return nullptr; }
    const unsigned char* q = p;
    p += n;
    return q;
  }
};
}  // namespace

extern "C" int ccvpe_ctx_create(const void* plan, long long n_bytes, void* weights_dev, void* workspace_dev, ccvpe_ctx** out) {
  if (!plan || n_bytes < 64 || !out) return fail(CCVPE_EINVAL, "ctx_create: null / short plan");
  Reader r{reinterpret_cast<const unsigned char*>(plan), reinterpret_cast<const unsigned char*>(plan) + n_bytes};
  const unsigned char* magic = r.bytes(8);
  if (!magic || std::memcmp(magic, "CCVPLAN1", 8)) return fail(CCVPE_EINVAL, "ctx_create: not a CCVPE plan (bad magic)");
  ccvpe_ctx* c = new ccvpe_ctx();
  auto bail = [&](int code, const char* msg) { delete c; return fail(code, "ctx_create: %s", msg); };
  const int abi = (int)r.get<uint32_t>();
  const uint32_t flags = r.get<uint32_t>();
  (void)flags;
  if (abi != ccvpe_abi_version()) {
    delete c;
    return fail(CCVPE_EINVAL, "ctx_create: the plan was recorded against ABI %d, this library is ABI %d", abi, ccvpe_abi_version());
  }
  c->workspace_bytes = r.get<uint64_t>();
  c->weights_bytes = r.get<uint64_t>();
  c->grd_bytes = r.get<uint64_t>();
  c->sat_bytes = r.get<uint64_t>();
  const uint32_t n_calls = r.get<uint32_t>(), n_blobs = r.get<uint32_t>(), n_out = r.get<uint32_t>();
  r.get<uint32_t>();
  if (!r.ok || n_calls > (1u << 20) || n_blobs > (1u << 20) || n_out > 64) return bail(CCVPE_EINVAL, "corrupt header");
  for (uint32_t i = 0; i < n_out; ++i) {
    Output o;
    o.off = r.get<uint64_t>(); o.bytes = r.get<uint64_t>(); o.dtype = r.get<uint32_t>(); o.ndim = r.get<uint32_t>();
    for (int k = 0; k < 4; ++k) o.dims[k] = (long long)r.get<uint64_t>();
    for (int k = 0; k < 4; ++k) o.strides[k] = (long long)r.get<uint64_t>();
    if (o.off > c->workspace_bytes || o.bytes > c->workspace_bytes - o.off) return bail(CCVPE_EINVAL, "output outside the workspace");   // (no uint64 wrap)
    c->outputs.push_back(o);
  }
  struct Reloc { uint32_t blob, field, kind; uint64_t value; };
  std::vector<Reloc> relocs;
  for (uint32_t i = 0; i < n_blobs && r.ok; ++i) {
    const uint32_t nb = r.get<uint32_t>(), nr = r.get<uint32_t>();
    for (uint32_t k = 0; k < nr; ++k) {
      Reloc q{i, r.get<uint32_t>(), r.get<uint32_t>(), r.get<uint64_t>()};
      if ((uint64_t)q.field + 8 > (uint64_t)nb) return bail(CCVPE_EINVAL, "relocation outside its blob");       // 64-bit: field = 0xFFFFFFFC must not wrap
      relocs.push_back(q);
    }
    const unsigned char* d = r.bytes((nb + 7) & ~7u);
    if (!d) return bail(CCVPE_EINVAL, "truncated blob");
    c->blobs.emplace_back(d, d + nb);
  }
  struct RawArg { uint32_t kind; uint64_t value; };
  std::vector<std::vector<RawArg>> raw(n_calls);
  for (uint32_t i = 0; i < n_calls && r.ok; ++i) {
    const unsigned char* nm = r.bytes(48);
    if (!nm) break;
    std::string name(reinterpret_cast<const char*>(nm), strnlen(reinterpret_cast<const char*>(nm), 48));
    const uint32_t na = r.get<uint32_t>();
    const uint32_t sid = r.get<uint32_t>() & 1u;
    if (na > 64) return bail(CCVPE_EINVAL, "call with more than 64 arguments");
    for (uint32_t k = 0; k < na; ++k) {
      RawArg a{r.get<uint32_t>(), 0};
      r.get<uint32_t>();
      a.value = r.get<uint64_t>();
      raw[i].push_back(a);
    }
    if (name == "@wait") {                             // stream raw[0] waits for what has been enqueued on stream raw[1]
      if (na != 2) return bail(CCVPE_EINVAL, "malformed stream wait");
      c->calls.push_back(Call{nullptr, name, 0, 0, (uint32_t)(raw[i][0].value & 1), (int)(raw[i][1].value & 1)});
      raw[i].clear();
      continue;
    }
    auto it = registry().find(name);
    if (it == registry().end()) {
      delete c;
      return fail(CCVPE_EINVAL, "ctx_create: the plan calls %s, which this library does not replay", name.c_str());
    }
    c->calls.push_back(Call{it->second, name, 0, na, sid, -1});
  }
  if (!r.ok) return bail(CCVPE_EINVAL, "truncated plan");
  // the weights blob: 256-byte aligned from the start of the plan
  size_t woff = (size_t)(r.p - reinterpret_cast<const unsigned char*>(plan));
  woff = (woff + 255) & ~(size_t)255;
  if (woff > (size_t)n_bytes || c->weights_bytes > (uint64_t)((size_t)n_bytes - woff)) return bail(CCVPE_EINVAL, "truncated weights");
  // device memory: the caller's (torch-owned in the Python binding) or the library's own
  if (weights_dev) c->weights = reinterpret_cast<char*>(weights_dev);
  else {
    if (hipMalloc(reinterpret_cast<void**>(&c->weights), c->weights_bytes ? c->weights_bytes : 256) != hipSuccess) return bail(CCVPE_ELAUNCH, "hipMalloc(weights)");
    c->own_weights = true;
  }
  if (workspace_dev) c->workspace = reinterpret_cast<char*>(workspace_dev);
  else {
    if (hipMalloc(reinterpret_cast<void**>(&c->workspace), c->workspace_bytes ? c->workspace_bytes : 256) != hipSuccess) {
      if (c->own_weights) (void)hipFree(c->weights);
      return bail(CCVPE_ELAUNCH, "hipMalloc(workspace)");
    }
    c->own_workspace = true;
  }
  if (c->weights_bytes &&
      hipMemcpy(c->weights, reinterpret_cast<const unsigned char*>(plan) + woff, c->weights_bytes, hipMemcpyHostToDevice) != hipSuccess) {
    ccvpe_ctx_destroy(c);
    return fail(CCVPE_ELAUNCH, "ctx_create: weight upload failed");
  }
  // resolve everything that does not change from call to call
  auto resolve = [&](uint32_t kind, uint64_t v, bool& dynamic, bool& bad) -> uint64_t {
    dynamic = bad = false;
    switch (kind) {
      case K_INT: case K_FLT: return v;
      case K_NULL: return 0;
      case K_WEIGHTS: bad = v >= c->weights_bytes; return (uint64_t)(uintptr_t)(c->weights + v);
      case K_WORKSPACE: bad = v >= c->workspace_bytes; return (uint64_t)(uintptr_t)(c->workspace + v);
      case K_BLOB: bad = v >= c->blobs.size(); return bad ? 0 : (uint64_t)(uintptr_t)c->blobs[v].data();
      case K_GRD: bad = v >= c->grd_bytes; dynamic = true; return 0;
      case K_SAT: bad = v >= c->sat_bytes; dynamic = true; return 0;
      case K_STREAM: dynamic = true; return 0;
      default: bad = true; return 0;
    }
  };
  for (const Reloc& q : relocs) {
    bool dyn, bad;
    const uint64_t v = resolve(q.kind, q.value, dyn, bad);
    if (bad || q.kind == K_BLOB || q.kind == K_STREAM) { ccvpe_ctx_destroy(c); return fail(CCVPE_EINVAL, "ctx_create: bad relocation"); }
    if (dyn) c->blob_patches.push_back(BlobPatch{q.blob, q.field, q.kind, q.value});
    else std::memcpy(c->blobs[q.blob].data() + q.field, &v, 8);
  }
  for (uint32_t i = 0; i < n_calls; ++i) {
    c->calls[i].first = (uint32_t)c->args.size();
    for (uint32_t k = 0; k < raw[i].size(); ++k) {
      bool dyn, bad;
      const uint64_t v = resolve(raw[i][k].kind, raw[i][k].value, dyn, bad);
      if (bad) { ccvpe_ctx_destroy(c); return fail(CCVPE_EINVAL, "ctx_create: bad argument %u of call %u (%s)", k, i, c->calls[i].name.c_str()); }
      if (dyn) c->patches.push_back(Patch{i, k, raw[i][k].kind, raw[i][k].value});
      c->args.push_back(v);
    }
  }
  bool two_streams = false;
  for (const Call& k : c->calls) {
    two_streams = two_streams || k.sid == 1 || !k.fn;
    if (!k.fn) {
      hipEvent_t ev;
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ccvpe_ctx_destroy(c); return fail(CCVPE_ELAUNCH, "ctx_create: hipEventCreate"); }
      c->events.push_back(ev);
    }
  }
  if (two_streams && hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) {
    ccvpe_ctx_destroy(c);
    return fail(CCVPE_ELAUNCH, "ctx_create: hipStreamCreate");
  }
  *out = c;
  return CCVPE_OK;
}

extern "C" int ccvpe_ctx_destroy(ccvpe_ctx* c) {
  if (!c) return CCVPE_OK;
  if (c->own_weights && c->weights) (void)hipFree(c->weights);
  if (c->own_workspace && c->workspace) (void)hipFree(c->workspace);
  for (hipEvent_t ev : c->events) (void)hipEventDestroy(ev);
  if (c->side) (void)hipStreamDestroy(c->side);
  delete c;
  return CCVPE_OK;
}

extern "C" int ccvpe_ctx_info(const ccvpe_ctx* c, long long* workspace_bytes, long long* weights_bytes, long long* grd_bytes,
                              long long* sat_bytes, int* n_outputs, int* n_calls) {
  if (!c) return fail(CCVPE_EINVAL, "ctx_info: null ctx");
  if (workspace_bytes) *workspace_bytes = (long long)c->workspace_bytes;
  if (weights_bytes) *weights_bytes = (long long)c->weights_bytes;
  if (grd_bytes) *grd_bytes = (long long)c->grd_bytes;
  if (sat_bytes) *sat_bytes = (long long)c->sat_bytes;
  if (n_outputs) *n_outputs = (int)c->outputs.size();
  if (n_calls) *n_calls = (int)c->calls.size();
  return CCVPE_OK;
}

extern "C" int ccvpe_ctx_output(const ccvpe_ctx* c, int i, void** dev_ptr, long long* bytes, int* ndim, long long* dims4,
                                long long* strides4) {
  if (!c || i < 0 || i >= (int)c->outputs.size()) return fail(CCVPE_EINVAL, "ctx_output: bad index");
  const Output& o = c->outputs[i];
  if (dev_ptr) *dev_ptr = c->workspace + o.off;
  if (bytes) *bytes = (long long)o.bytes;
  if (ndim) *ndim = (int)o.ndim;
  if (dims4) for (int k = 0; k < 4; ++k) dims4[k] = o.dims[k];
  if (strides4) for (int k = 0; k < 4; ++k) strides4[k] = o.strides[k];
  return CCVPE_OK;
}

// One forward: grd [B,3,h,w], sat [B,3,H,W] NCHW fp32 device tensors of the recorded shapes; every kernel is enqueued on
// `stream`; out (optional) receives the device pointers of the nine outputs (logits, heatmap, ori, score1..6 — fp32, in the
// ctx's workspace: valid until the next ccvpe_forward on this ctx).  Not re-entrant per ctx (one ctx per stream).
extern "C" int ccvpe_forward(ccvpe_ctx* c, const void* grd, const void* sat, void** out, void* stream) {
  if (!c || !grd || !sat) return fail(CCVPE_EINVAL, "forward: null ctx / input");
  for (const Patch& q : c->patches) {
    uint64_t v = 0;
    if (q.kind == K_GRD) v = (uint64_t)(uintptr_t)(reinterpret_cast<const char*>(grd) + q.off);
    else if (q.kind == K_SAT) v = (uint64_t)(uintptr_t)(reinterpret_cast<const char*>(sat) + q.off);
    else v = (uint64_t)(uintptr_t)(c->calls[q.call].sid ? (void*)c->side : stream);
    c->args[c->calls[q.call].first + q.arg] = v;
  }
  for (const BlobPatch& q : c->blob_patches) {
    const uint64_t v = (uint64_t)(uintptr_t)(reinterpret_cast<const char*>(q.kind == K_GRD ? grd : sat) + q.off);
    std::memcpy(c->blobs[q.blob].data() + q.field, &v, 8);
  }
  size_t nev = 0;
  for (const Call& k : c->calls) {
    if (!k.fn) {                                        // fork / join: stream k.sid waits for stream k.wait_on
      hipStream_t waiter = k.sid ? c->side : (hipStream_t)stream, waitee = k.wait_on ? c->side : (hipStream_t)stream;
      hipEvent_t ev = c->events[nev++];
      if (hipEventRecord(ev, waitee) != hipSuccess || hipStreamWaitEvent(waiter, ev, 0) != hipSuccess)
        return fail(CCVPE_ELAUNCH, "forward: stream wait failed");
      continue;
    }
    const int rc = k.fn(c->args.data() + k.first, (int)k.nargs);
    if (rc != CCVPE_OK) return rc;                      // (the failing entry point has set ccvpe_last_error)
  }
  if (out) for (size_t i = 0; i < c->outputs.size(); ++i) out[i] = c->workspace + c->outputs[i].off;
  return CCVPE_OK;
}
