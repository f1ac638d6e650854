// gpu_stub.cc — TEST SCAFFOLDING, not product code.  The host side of libfmx.so (fmx_host.cc, fmx_text.cc, fmx_capi.cc: the .fmi
// reader / writer, the index builder, the text front-end, the C ABI's argument handling) is compiled with
// -fsanitize=address,undefined on the CPU box (GPU sanitizers are not available); this file stands in for fmx_gpu.hip there:
// every device entry point reports "no device", like libfmx.so itself does on a box without a gfx950 GPU.  Nothing here
// computes anything.
#include "fmx_gpu.hh"

namespace fmx {

struct DeviceIndex::Impl {};
struct DeviceIndex::Pending {};
static void no_device() { throw GpuError("sanitizer build: no device code"); }

int   gpu_device_count() { return 0; }
void  gpu_sort_resident_free(SortResident* r) { delete r; }
SortResident* gpu_suffix_sort(HostIndex&, int) { no_device(); return nullptr; }
void* gpu_host_alloc(size_t) { return nullptr; }
void  gpu_host_free(void*) {}
const char* stage_name(int) { return ""; }

DeviceIndex::DeviceIndex() : impl_(nullptr) {}
DeviceIndex::~DeviceIndex() {}
void     DeviceIndex::upload(const HostIndex&, int, SortResident*) { no_device(); }
int      DeviceIndex::device() const { return -1; }
uint64_t DeviceIndex::download(int, void*, uint64_t) const { no_device(); return 0; }
uint64_t DeviceIndex::bytes() const { return 0; }
uint32_t DeviceIndex::sub_queries() const { return 1; }
void DeviceIndex::set_sub_queries(uint32_t) {}
void DeviceIndex::set_host_sub_queries(uint32_t) {}
void DeviceIndex::set_initial_segment(uint64_t) {}
void DeviceIndex::set_heavy_min(uint32_t) {}
void DeviceIndex::set_probe_count(bool) {}
void DeviceIndex::set_serial(bool) {}
void DeviceIndex::set_turns(uint32_t) {}
void DeviceIndex::set_device_turns(uint32_t) {}
void DeviceIndex::set_expand_order(uint32_t) {}
void DeviceIndex::set_b_grid(uint32_t) {}
void DeviceIndex::set_heavy_grid(uint32_t) {}
void DeviceIndex::set_light_max(uint32_t) {}
void DeviceIndex::set_light_flat(uint32_t) {}
void DeviceIndex::set_probe_split(bool) {}
void DeviceIndex::set_contexts(uint32_t) {}
void DeviceIndex::set_max_host_active(uint32_t) {}
void DeviceIndex::set_pair_finish(bool) {}
void DeviceIndex::set_small_path(bool) {}
void DeviceIndex::set_real_hybrid(bool) {}
void DeviceIndex::match_device(const HostIndex&, const Params&, const DeviceBatchView&, Rec*, uint32_t*, uint32_t*, uint32_t, void*,
                               const std::vector<uint32_t>&, Stats*) { no_device(); }
DeviceIndex::Pending* DeviceIndex::match_device_begin(const HostIndex&, const Params&, const DeviceBatchView&, Rec*, uint32_t*, uint32_t*,
                                                      uint32_t, void*, const std::vector<uint32_t>&) { no_device(); return nullptr; }
void DeviceIndex::match_device_end(Pending*) { no_device(); }
void DeviceIndex::match_device_abandon(Pending*) {}
void DeviceIndex::match_host(const HostIndex&, const Params&, const HostBatchView&, Rec*, uint32_t*, uint32_t*, uint32_t, Stats*, Rec*,
                             uint32_t*) { no_device(); }
struct DeviceIndex::HostCall {};
DeviceIndex::HostCall* DeviceIndex::match_host_begin(const HostIndex&, const Params&, const HostBatchView&, Rec*, uint32_t*, uint32_t*, uint32_t, Rec*,
                                                     uint32_t*) { no_device(); return nullptr; }
void DeviceIndex::match_host_end(HostCall*) { no_device(); }
void DeviceIndex::match_host_abandon(HostCall*) {}
void DeviceIndex::match_lines(const HostIndex&, const Params&, const char*, uint64_t, uint64_t, Rec*, uint32_t*, uint32_t*, uint32_t, Stats*) { no_device(); }
void DeviceIndex::match_raw_lines(const HostIndex&, const Params&, const char*, uint64_t, uint64_t, Rec*, uint32_t*, uint32_t*, uint32_t, Stats*,
                                  const RawLineTok&, std::vector<uint64_t>&) { no_device(); }
void DeviceIndex::subsequence_host(const HostIndex&, const HostBatchView&, uint32_t, bool, int, float, bool, Rec*, uint32_t*, uint32_t*) { no_device(); }
std::vector<uint32_t> DeviceIndex::sub_bounds_from_device(const uint32_t*, uint64_t, uint32_t*, void*) { no_device(); return {}; }

}  // namespace fmx
