// tt_abi_version / tt_last_error and the error plumbing shared by every entry point.
#include "common.h"
#include <mutex>
#include <string>
#include <vector>
#include <cstring>
#include <cstdlib>

namespace tt {
static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ---- kernel timing --------------------------------------------------------------------------
bool g_prof_on = false;
int g_prof_stride = 1;
namespace {
struct ProfTag {
  std::string name;
  std::vector<hipEvent_t> ev;   // 2 per launch: begin, end
  int used = 0;                 // launches recorded
  int seen = 0;                 // launches seen (may exceed capacity)
  bool skip = false;            // the begin of the launch in flight was not recorded (stride / capacity)
};
std::mutex g_prof_mu;
std::vector<ProfTag> g_prof_tags;
ProfTag* find_tag(const char* tag) {
  for (auto& t : g_prof_tags)
    if (t.name == tag) return &t;
  return nullptr;
}
}  // namespace

void prof_record(const char* tag, hipStream_t stream, bool end) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfTag* t = find_tag(tag);
  if (t == nullptr) return;
  if (!end) {
    // every hipEventRecord is a barrier packet on the stream (4-7 us): with a stride only every n-th launch pays it
    t->skip = (g_prof_stride > 1 && t->seen % g_prof_stride != 0) || 2 * (t->used + 1) > (int)t->ev.size();
    if (t->skip) { ++t->seen; return; }
    (void)hipEventRecord(t->ev[2 * t->used], stream);
  } else {
    if (t->skip) { t->skip = false; return; }
    (void)hipEventRecord(t->ev[2 * t->used + 1], stream);
    ++t->used;
    ++t->seen;
  }
}

// one kernel under a tag (tt::launch): the event pair the runtime fills with the dispatch's own begin / end timestamps
bool prof_kernel_events(const char* tag, hipStream_t stream, hipEvent_t* start, hipEvent_t* stop, bool* bracket) {
  static const bool brackets = std::getenv("TT_PROF_BRACKETS") != nullptr && std::atoi(std::getenv("TT_PROF_BRACKETS")) != 0;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  *bracket = false;
  ProfTag* t = find_tag(tag);
  if (t == nullptr) return false;
  const bool skip = (g_prof_stride > 1 && t->seen % g_prof_stride != 0) || 2 * (t->used + 1) > (int)t->ev.size();
  if (skip) { ++t->seen; return false; }
  if (brackets) {
    (void)hipEventRecord(t->ev[2 * t->used], stream);
    *bracket = true;
    return false;
  }
  *start = t->ev[2 * t->used];
  *stop = t->ev[2 * t->used + 1];
  ++t->used;
  ++t->seen;
  return true;
}
void prof_kernel_end(const char* tag, hipStream_t stream) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfTag* t = find_tag(tag);
  if (t == nullptr) return;
  (void)hipEventRecord(t->ev[2 * t->used + 1], stream);
  ++t->used;
  ++t->seen;
}
}  // namespace tt

extern "C" int tt_profile_set_stride(int32_t stride) {
  if (stride < 1) return tt::fail(TT_ERR_INVALID_ARG, "tt_profile_set_stride: stride must be >= 1");
  std::lock_guard<std::mutex> lk(tt::g_prof_mu);
  tt::g_prof_stride = stride;
  return TT_OK;
}

extern "C" int tt_profile_enable(const char* tags_csv, int32_t capacity_per_tag) {
  std::lock_guard<std::mutex> lk(tt::g_prof_mu);
  for (auto& t : tt::g_prof_tags)
    for (auto e : t.ev) (void)hipEventDestroy(e);
  tt::g_prof_tags.clear();
  tt::g_prof_on = false;
  if (tags_csv == nullptr || tags_csv[0] == 0) return TT_OK;
  if (capacity_per_tag <= 0) return tt::fail(TT_ERR_INVALID_ARG, "tt_profile_enable: capacity must be positive");
  std::string s(tags_csv);
  size_t pos = 0;
  while (pos <= s.size()) {
    size_t q = s.find(',', pos);
    if (q == std::string::npos) q = s.size();
    if (q > pos) {
      tt::ProfTag t;
      t.name = s.substr(pos, q - pos);
      t.ev.resize(2 * (size_t)capacity_per_tag);
      for (auto& e : t.ev)
        if (hipEventCreate(&e) != hipSuccess) return tt::fail(TT_ERR_LAUNCH, "tt_profile_enable: hipEventCreate failed");
      tt::g_prof_tags.push_back(std::move(t));
    }
    pos = q + 1;
  }
  tt::g_prof_on = !tt::g_prof_tags.empty();
  return TT_OK;
}

extern "C" int tt_profile_read(const char* tag, float* ms, int32_t cap, int32_t* count) {
  if (tag == nullptr || count == nullptr || (cap > 0 && ms == nullptr))
    return tt::fail(TT_ERR_INVALID_ARG, "tt_profile_read: null pointer");
  std::lock_guard<std::mutex> lk(tt::g_prof_mu);
  tt::ProfTag* t = tt::find_tag(tag);
  if (t == nullptr) return tt::fail(TT_ERR_INVALID_ARG, "tt_profile_read: tag '%s' is not enabled", tag);
  *count = t->used < cap ? t->used : cap;
  for (int i = 0; i < t->used && i < cap; ++i) {
    if (hipEventSynchronize(t->ev[2 * i + 1]) != hipSuccess ||
        hipEventElapsedTime(&ms[i], t->ev[2 * i], t->ev[2 * i + 1]) != hipSuccess)
      return tt::fail(TT_ERR_LAUNCH, "tt_profile_read: event query failed for '%s'", tag);
  }
  t->used = 0;
  t->seen = 0;
  return TT_OK;
}

extern "C" int tt_abi_version(void) { return TT_ABI_VERSION; }
// sizeof of the ABI's structs, so that a binding can check its mirrors (which = 0 tt_train_step, 1 tt_dense_fwd_args,
// 2 tt_dense_bwd_args, 3 tt_sparse_table_ids, 4 tt_dense_seg, 5 tt_id_buckets, 6 tt_dense_lookup; anything else: -1)
extern "C" int64_t tt_abi_struct_bytes(int32_t which) {
  switch (which) {
    case 0: return (int64_t)sizeof(tt_train_step);
    case 1: return (int64_t)sizeof(tt_dense_fwd_args);
    case 2: return (int64_t)sizeof(tt_dense_bwd_args);
    case 3: return (int64_t)sizeof(tt_sparse_table_ids);
    case 4: return (int64_t)sizeof(tt_dense_seg);
    case 5: return (int64_t)sizeof(tt_id_buckets);
    case 6: return (int64_t)sizeof(tt_dense_lookup);
    default: return -1;
  }
}
extern "C" const char* tt_last_error(void) { return tt::err_buf(); }
