// A caller WITHOUT Python: loads a plan file written by ccvpe_amd/plan.py (Plan.save), creates a ccvpe_ctx, runs the forward
// on synthetic (or file-provided) inputs and writes the nine outputs to a raw file — tests/test_plan_gpu.py compares them with
// the Python forward.  This is the binding a C / C++ / Go (cgo) / Rust (FFI) host would write against include/ccvpe_hip.h.
//   hipcc -O2 -I include tools/plan_run.cpp -L ccvpe_amd -lccvpe_hip -Wl,-rpath,$PWD/ccvpe_amd -o plan_run
//   ./plan_run model.plan grd.f32 sat.f32 out.f32 [reps]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ccvpe_hip.h"

#define HIP_OK(call)                                                                  \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); exit(1); } \
  } while (0)

static std::vector<char> slurp(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<char> b((size_t)n);
  if (fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read on %s\n", path); exit(2); }
  fclose(f);
  return b;
}

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s plan grd.f32 sat.f32 out.f32 [reps]\n", argv[0]); return 2; }
  const int reps = argc > 5 ? atoi(argv[5]) : 1;
  std::vector<char> plan = slurp(argv[1]), grd = slurp(argv[2]), sat = slurp(argv[3]);
  ccvpe_ctx* ctx = nullptr;
  if (ccvpe_ctx_create(plan.data(), (long long)plan.size(), nullptr, nullptr, &ctx)) {   // the library owns weights + workspace
    fprintf(stderr, "ccvpe_ctx_create: %s\n", ccvpe_last_error());
    return 1;
  }
  long long ws, wb, gb, sb;
  int nout, ncalls;
  ccvpe_ctx_info(ctx, &ws, &wb, &gb, &sb, &nout, &ncalls);
  if ((long long)grd.size() != gb || (long long)sat.size() != sb) {
    fprintf(stderr, "input sizes %zu / %zu do not match the plan (%lld / %lld bytes)\n", grd.size(), sat.size(), gb, sb);
    return 1;
  }
  void *dg = nullptr, *ds = nullptr;
  HIP_OK(hipMalloc(&dg, grd.size()));
  HIP_OK(hipMalloc(&ds, sat.size()));
  HIP_OK(hipMemcpy(dg, grd.data(), grd.size(), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(ds, sat.data(), sat.size(), hipMemcpyHostToDevice));
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  void* outs[16] = {nullptr};
  for (int r = 0; r < 2; ++r)                                   // warm-up
    if (ccvpe_forward(ctx, dg, ds, outs, st)) { fprintf(stderr, "ccvpe_forward: %s\n", ccvpe_last_error()); return 1; }
  HIP_OK(hipStreamSynchronize(st));
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r)
    if (ccvpe_forward(ctx, dg, ds, outs, st)) { fprintf(stderr, "ccvpe_forward: %s\n", ccvpe_last_error()); return 1; }
  HIP_OK(hipStreamSynchronize(st));
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
  FILE* f = fopen(argv[4], "wb");
  for (int i = 0; i < nout; ++i) {
    void* p;
    long long bytes, d[4], s[4];
    int nd;
    ccvpe_ctx_output(ctx, i, &p, &bytes, &nd, d, s);
    if (p != outs[i]) { fprintf(stderr, "output %d: ccvpe_forward and ccvpe_ctx_output disagree\n", i); return 1; }
    std::vector<float> h((size_t)bytes / 4);
    HIP_OK(hipMemcpy(h.data(), p, (size_t)bytes, hipMemcpyDeviceToHost));
    for (long long a = 0; a < d[0]; ++a)               // written densely, in index order (an output may be a strided view)
      for (long long b = 0; b < d[1]; ++b)
        for (long long c = 0; c < d[2]; ++c)
          for (long long e = 0; e < d[3]; ++e) fwrite(&h[(size_t)(a * s[0] + b * s[1] + c * s[2] + e * s[3])], 4, 1, f);
  }
  fclose(f);
  printf("{\"calls\": %d, \"outputs\": %d, \"workspace_mib\": %.1f, \"weights_mib\": %.1f, \"ms_per_forward\": %.3f}\n", ncalls, nout,
         ws / 1048576.0, wb / 1048576.0, ms);
  ccvpe_ctx_destroy(ctx);
  return 0;
}
