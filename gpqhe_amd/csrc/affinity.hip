// affinity.hip -- host placement for C hosts that drive several GPUs from one process (tests/c/shard_host.c: one worker thread per shard).
//
// The rank processes of bench.py bind themselves in Python before their first HIP call (gpqhe_amd/affinity.py); a C host does the same per
// worker THREAD with gpq_bind_thread_to_device(device) as the thread's first action -- before gpq_set_device, so that the page-locked buffers
// it allocates next are first-touched on the GPU's own socket and its launches do not cross the socket interconnect.  Pure host code, no HIP
// call (sysfs only), same mapping as the Python side:
//   HIP device i -> i-th GPU node of /sys/class/kfd/kfd/topology/nodes (simd_count > 0, /dev/dri/renderD<minor> openable), after
//   ROCR_VISIBLE_DEVICES, then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (integer lists) -> /sys/class/drm/renderD<minor>/device/local_cpulist.
// "Cannot tell" is never an error: the functions return 0 and the thread stays where it was.
#include <dirent.h>
#include <pthread.h>
#include <sched.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gpqhe_hip.h"

namespace {

struct GpuNode { int node, minor; };

bool read_text(const std::string &path, std::string *out) {
  FILE *f = fopen(path.c_str(), "r");
  if (!f) return false;
  char buf[4096];
  out->clear();
  size_t k;
  while ((k = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, k);
  fclose(f);
  return true;
}

long prop(const std::string &text, const char *key, long dflt) {
  size_t pos = 0;
  const size_t kl = strlen(key);
  while (pos < text.size()) {
    size_t eol = text.find('\n', pos);
    if (eol == std::string::npos) eol = text.size();
    if (text.compare(pos, kl, key) == 0 && pos + kl < eol && text[pos + kl] == ' ') return strtol(text.c_str() + pos + kl + 1, nullptr, 10);
    pos = eol + 1;
  }
  return dflt;
}

std::vector<GpuNode> gpu_nodes(const std::string &root) {
  std::vector<GpuNode> out;
  const std::string base = root + "sys/class/kfd/kfd/topology/nodes";
  DIR *d = opendir(base.c_str());
  if (!d) return out;
  std::vector<int> ids;
  while (dirent *e = readdir(d)) {
    char *end = nullptr;
    const long v = strtol(e->d_name, &end, 10);
    if (end != e->d_name && *end == 0) ids.push_back((int)v);
  }
  closedir(d);
  std::sort(ids.begin(), ids.end());
  for (int id : ids) {
    std::string text;
    if (!read_text(base + "/" + std::to_string(id) + "/properties", &text)) continue;
    if (prop(text, "simd_count", 0) <= 0) continue;
    const long minor = prop(text, "drm_render_minor", -1);
    if (minor < 0) continue;
    if (access((root + "dev/dri/renderD" + std::to_string(minor)).c_str(), R_OK | W_OK) != 0) continue;
    out.push_back({id, (int)minor});
  }
  return out;
}

// integer list of a *_VISIBLE_DEVICES variable; false on anything else (UUID forms): the caller gives up
bool int_list(const char *text, std::vector<int> *out) {
  out->clear();
  const char *p = text;
  while (*p) {
    while (*p == ' ' || *p == ',') ++p;
    if (!*p) break;
    char *end = nullptr;
    const long v = strtol(p, &end, 10);
    if (end == p) return false;
    while (*end == ' ') ++end;
    if (*end && *end != ',') return false;
    out->push_back((int)v);
    p = end;
  }
  return true;
}

bool apply_visible(std::vector<GpuNode> *nodes) {
  const char *layers[2][2] = {{"ROCR_VISIBLE_DEVICES", nullptr}, {"HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"}};
  for (auto &layer : layers) {
    const char *text = nullptr;
    for (const char *name : layer)
      if (name && !text) { const char *v = getenv(name); if (v && *v) text = v; }
    if (!text) continue;
    std::vector<int> idx;
    if (!int_list(text, &idx)) return false;
    std::vector<GpuNode> picked;
    for (int i : idx) {
      if (i < 0 || i >= (int)nodes->size()) break;      // the runtimes stop at the first invalid index
      picked.push_back((*nodes)[i]);
    }
    nodes->swap(picked);
  }
  return true;
}

// "0-3,8" -> cpu_set_t; returns the count
int parse_cpulist(const std::string &text, cpu_set_t *set) {
  CPU_ZERO(set);
  int count = 0;
  const char *p = text.c_str();
  while (*p) {
    while (*p == ',' || *p == ' ' || *p == '\n') ++p;
    if (!*p) break;
    char *end = nullptr;
    long lo = strtol(p, &end, 10), hi = lo;
    if (end == p) return 0;
    if (*end == '-') { p = end + 1; hi = strtol(p, &end, 10); if (end == p) return 0; }
    for (long c = lo; c <= hi && c < CPU_SETSIZE; ++c) if (c >= 0 && !CPU_ISSET(c, set)) { CPU_SET(c, set); ++count; }
    p = end;
  }
  return count;
}

int local_cpus(int device, const char *sysfs_root, std::string *list, cpu_set_t *set) {
  std::string root = sysfs_root && *sysfs_root ? sysfs_root : "/";
  if (root.back() != '/') root += '/';
  std::vector<GpuNode> nodes = gpu_nodes(root);
  if (!apply_visible(&nodes)) return 0;
  if (device < 0 || device >= (int)nodes.size()) return 0;
  const std::string dev = root + "sys/class/drm/renderD" + std::to_string(nodes[device].minor) + "/device/";
  std::string numa;
  if (!read_text(dev + "numa_node", &numa) || strtol(numa.c_str(), nullptr, 10) < 0) return 0;
  if (!read_text(dev + "local_cpulist", list)) return 0;
  while (!list->empty() && (list->back() == '\n' || list->back() == ' ')) list->pop_back();
  return parse_cpulist(*list, set);
}

}  // namespace

extern "C" int gpq_device_local_cpus(int device, const char *sysfs_root, char *cpulist, size_t cap) {
  std::string list;
  cpu_set_t set;
  const int count = local_cpus(device, sysfs_root, &list, &set);
  if (cpulist && cap) snprintf(cpulist, cap, "%s", count ? list.c_str() : "");
  return count;
}

extern "C" int gpq_bind_thread_to_device(int device) {
  std::string list;
  cpu_set_t want, have;
  if (!local_cpus(device, nullptr, &list, &want)) return 0;
  if (pthread_getaffinity_np(pthread_self(), sizeof have, &have) != 0) return 0;
  cpu_set_t both;
  CPU_AND(&both, &want, &have);
  const int count = CPU_COUNT(&both);
  if (!count || CPU_EQUAL(&both, &have)) return 0;               // nothing allowed on that node, or already confined to it
  return pthread_setaffinity_np(pthread_self(), sizeof both, &both) == 0 ? count : 0;
}
