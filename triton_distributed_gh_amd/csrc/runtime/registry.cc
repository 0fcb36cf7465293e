// Native op registry + AOT kernel table (host side, no GPU needed).
//
// Role in the reference: csrc/lib/registry.{h,cc} (`OpInitRegistry` singleton, register_one /
// initialize_all) and the generated AOT dispatch tables of tools/compile_aot.py:357-435
// (`distributed.<kernel>(stream, args..., algo_info)` with a `<kernel>__triton_algo_info_t` struct).
//
// Every AOT kernel entry is (name, signature string, grid expression triple, list of algo-infos);
// an algo-info is a small set of named integer constants (tile sizes, waves, stages ...) plus the
// exported launcher symbol that implements it.  Entries are registered at build time from the
// kernel specs (tools/compile_aot.py writes them into a generated .cc that calls
// tdh_registry_add_*), looked up by name at run time and dispatched by the best-matching
// algo-info.  The registry itself is plain C++ so it can be unit-tested on a CPU-only machine.
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#define TDH_API extern "C" __attribute__((visibility("default")))

namespace {

struct AlgoInfo {
  std::vector<std::pair<std::string, int64_t>> consts;   // e.g. BLOCK_N=64, num_warps=4
  std::string launcher;                                   // exported symbol that runs this variant
};

struct KernelEntry {
  std::string name, signature, grid[3];
  std::vector<AlgoInfo> algos;
};

struct InitFn {
  std::string name;
  void (*fn)();
  bool done = false;
};

class Registry {
 public:
  static Registry& get() {
    static Registry r;
    return r;
  }
  std::mutex mu;
  std::map<std::string, KernelEntry> kernels;
  std::vector<std::string> order;
  std::vector<InitFn> inits;
};

thread_local std::string g_scratch;

}  // namespace

// ---- op initialisers ("OpInitRegistry::register_one / initialize_all") --------------------------------
TDH_API int tdh_registry_register_init(const char* name, void (*fn)()) {
  auto& r = Registry::get();
  std::lock_guard<std::mutex> g(r.mu);
  for (auto& i : r.inits)
    if (i.name == name) return 1;
  r.inits.push_back({name, fn, false});
  return 0;
}

TDH_API int tdh_registry_initialize_all() {
  auto& r = Registry::get();
  std::vector<InitFn*> todo;
  {
    std::lock_guard<std::mutex> g(r.mu);
    for (auto& i : r.inits)
      if (!i.done) {
        i.done = true;
        todo.push_back(&i);
      }
  }
  for (auto* i : todo)
    if (i->fn) i->fn();
  return (int)todo.size();
}

// ---- AOT kernel table ----------------------------------------------------------------------------
TDH_API int tdh_registry_add_kernel(const char* name, const char* signature, const char* gx, const char* gy,
                                    const char* gz) {
  auto& r = Registry::get();
  std::lock_guard<std::mutex> g(r.mu);
  auto it = r.kernels.find(name);
  if (it == r.kernels.end()) {
    r.order.push_back(name);
    it = r.kernels.emplace(name, KernelEntry{}).first;
  }
  KernelEntry& k = it->second;
  k.name = name;
  k.signature = signature ? signature : "";
  k.grid[0] = gx ? gx : "1";
  k.grid[1] = gy ? gy : "1";
  k.grid[2] = gz ? gz : "1";
  return 0;
}

// consts: "BLOCK_N=64;BLOCK_H=16;num_warps=4"
TDH_API int tdh_registry_add_algo(const char* kernel, const char* consts, const char* launcher) {
  auto& r = Registry::get();
  std::lock_guard<std::mutex> g(r.mu);
  auto it = r.kernels.find(kernel);
  if (it == r.kernels.end()) return -1;
  AlgoInfo a;
  a.launcher = launcher ? launcher : "";
  std::string s = consts ? consts : "";
  size_t pos = 0;
  while (pos < s.size()) {
    size_t end = s.find(';', pos);
    if (end == std::string::npos) end = s.size();
    const std::string item = s.substr(pos, end - pos);
    const size_t eq = item.find('=');
    if (eq != std::string::npos) a.consts.emplace_back(item.substr(0, eq), std::stoll(item.substr(eq + 1)));
    pos = end + 1;
  }
  it->second.algos.push_back(std::move(a));
  return (int)it->second.algos.size() - 1;
}

TDH_API int tdh_registry_num_kernels() {
  auto& r = Registry::get();
  std::lock_guard<std::mutex> g(r.mu);
  return (int)r.order.size();
}

TDH_API const char* tdh_registry_kernel_name(int i) {
  auto& r = Registry::get();
  std::lock_guard<std::mutex> g(r.mu);
  if (i < 0 || i >= (int)r.order.size()) return nullptr;
  g_scratch = r.order[i];
  return g_scratch.c_str();
}

TDH_API const char* tdh_registry_signature(const char* kernel) {
  auto& r = Registry::get();
  std::lock_guard<std::mutex> g(r.mu);
  auto it = r.kernels.find(kernel);
  if (it == r.kernels.end()) return nullptr;
  g_scratch = it->second.signature;
  return g_scratch.c_str();
}

TDH_API const char* tdh_registry_grid(const char* kernel, int axis) {
  auto& r = Registry::get();
  std::lock_guard<std::mutex> g(r.mu);
  auto it = r.kernels.find(kernel);
  if (it == r.kernels.end() || axis < 0 || axis > 2) return nullptr;
  g_scratch = it->second.grid[axis];
  return g_scratch.c_str();
}

TDH_API int tdh_registry_num_algos(const char* kernel) {
  auto& r = Registry::get();
  std::lock_guard<std::mutex> g(r.mu);
  auto it = r.kernels.find(kernel);
  return it == r.kernels.end() ? -1 : (int)it->second.algos.size();
}

// Serialised form of algo `idx`: "launcher|K1=V1;K2=V2".
TDH_API const char* tdh_registry_algo(const char* kernel, int idx) {
  auto& r = Registry::get();
  std::lock_guard<std::mutex> g(r.mu);
  auto it = r.kernels.find(kernel);
  if (it == r.kernels.end() || idx < 0 || idx >= (int)it->second.algos.size()) return nullptr;
  const AlgoInfo& a = it->second.algos[idx];
  g_scratch = a.launcher + "|";
  for (size_t i = 0; i < a.consts.size(); ++i) {
    if (i) g_scratch += ";";
    g_scratch += a.consts[i].first + "=" + std::to_string(a.consts[i].second);
  }
  return g_scratch.c_str();
}

// Pick the algo whose constants match the request: every requested (key, value) present in the
// algo must be equal; among matches the one agreeing on most keys wins.  Returns -1 if none.
TDH_API int tdh_registry_select_algo(const char* kernel, const char* want) {
  auto& r = Registry::get();
  std::lock_guard<std::mutex> g(r.mu);
  auto it = r.kernels.find(kernel);
  if (it == r.kernels.end()) return -1;
  std::vector<std::pair<std::string, int64_t>> req;
  std::string s = want ? want : "";
  size_t pos = 0;
  while (pos < s.size()) {
    size_t end = s.find(';', pos);
    if (end == std::string::npos) end = s.size();
    const std::string item = s.substr(pos, end - pos);
    const size_t eq = item.find('=');
    if (eq != std::string::npos) req.emplace_back(item.substr(0, eq), std::stoll(item.substr(eq + 1)));
    pos = end + 1;
  }
  int best = -1, best_score = -1;
  for (size_t i = 0; i < it->second.algos.size(); ++i) {
    int score = 0;
    bool ok = true;
    for (auto& q : req)
      for (auto& c : it->second.algos[i].consts)
        if (c.first == q.first) {
          if (c.second == q.second)
            ++score;
          else
            ok = false;
        }
    if (ok && score > best_score) best = (int)i, best_score = score;
  }
  return best;
}

TDH_API void tdh_registry_clear() {
  auto& r = Registry::get();
  std::lock_guard<std::mutex> g(r.mu);
  r.kernels.clear();
  r.order.clear();
}
