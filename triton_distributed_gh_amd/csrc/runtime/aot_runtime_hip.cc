// Run-time loader for ahead-of-time compiled code objects (HSACO): per-device module cache,
// function lookup and launch.
//
// Role in the reference: tools/runtime/triton_aot_runtime.{cc,h} (`CUDAModule` per-context module
// cache :54-97, `CUDAFunction` :99-135, C API CUDAModuleLoadData / GetFunction / Unload /
// FuncSetAttribute / LaunchKernel :140-199) and the generated launchers of tools/compile/compile.c.
// The in-tree kernels are linked into libtdist_hip.so and launched directly; this loader serves
// code objects produced separately (tools/compile_aot.py --build emits .hsaco files with hipcc
// --genco) so user kernels can join the registry without relinking the library.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "tdist/common.hpp"

namespace {

struct ModuleKey {
  const void* image;
  int device;
  bool operator<(const ModuleKey& o) const { return image != o.image ? image < o.image : device < o.device; }
};

std::mutex g_mu;
std::map<ModuleKey, hipModule_t> g_modules;                       // (image, device) -> loaded module
std::map<std::pair<hipModule_t, std::string>, hipFunction_t> g_functions;

}  // namespace

// Load `image` (an HSACO blob that stays alive in the caller) on the current device, once.
TDIST_API int tdist_aot_module_load_data(const void* image, void** module_out) {
  int dev = 0;
  TDIST_HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> g(g_mu);
  const ModuleKey key{image, dev};
  auto it = g_modules.find(key);
  if (it == g_modules.end()) {
    hipModule_t m;
    TDIST_HIP_TRY(hipModuleLoadData(&m, image));
    it = g_modules.emplace(key, m).first;
  }
  *module_out = (void*)it->second;
  return 0;
}

TDIST_API int tdist_aot_module_load_file(const char* path, void** module_out) {
  hipModule_t m;
  TDIST_HIP_TRY(hipModuleLoad(&m, path));
  *module_out = (void*)m;
  return 0;
}

TDIST_API int tdist_aot_module_get_function(void* module, const char* name, void** fn_out) {
  std::lock_guard<std::mutex> g(g_mu);
  const auto key = std::make_pair((hipModule_t)module, std::string(name));
  auto it = g_functions.find(key);
  if (it == g_functions.end()) {
    hipFunction_t f;
    TDIST_HIP_TRY(hipModuleGetFunction(&f, (hipModule_t)module, name));
    it = g_functions.emplace(key, f).first;
  }
  *fn_out = (void*)it->second;
  return 0;
}

TDIST_API int tdist_aot_module_unload(void* module) {
  std::lock_guard<std::mutex> g(g_mu);
  for (auto it = g_functions.begin(); it != g_functions.end();)
    it = it->first.first == (hipModule_t)module ? g_functions.erase(it) : std::next(it);
  for (auto it = g_modules.begin(); it != g_modules.end();)
    it = it->second == (hipModule_t)module ? g_modules.erase(it) : std::next(it);
  TDIST_HIP_TRY(hipModuleUnload((hipModule_t)module));
  return 0;
}

TDIST_API int tdist_aot_func_set_max_dynamic_lds(void* fn, int bytes) {
  TDIST_HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return 0;
}

// Workgroups of `block_threads` threads and `lds_bytes` of dynamic LDS the device keeps resident per compute
// unit for this kernel (what a persistent or grid-synchronising user kernel sizes its grid by).
TDIST_API int tdist_aot_occupancy(void* fn, int block_threads, int lds_bytes, int* out) {
  TDIST_HIP_TRY(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(out, (hipFunction_t)fn, block_threads, (size_t)lds_bytes));
  return 0;
}

// args: array of pointers to the kernel arguments (hipModuleLaunchKernel convention).
TDIST_API int tdist_aot_launch(void* fn, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz,
                               unsigned lds_bytes, void* stream, void** args) {
  TDIST_REQUIRE(gx > 0 && gy > 0 && gz > 0 && bx > 0 && by > 0 && bz > 0, "empty launch");
  TDIST_HIP_TRY(hipModuleLaunchKernel((hipFunction_t)fn, gx, gy, gz, bx, by, bz, lds_bytes, (hipStream_t)stream, args,
                                      nullptr));
  return 0;
}
