// hip_abi.hip -- host side of the C ABI declared in include/cloudini_hip.h.
//
// Plan building restates the reference's encoder selection (src/codec_common.cpp:69-153,
// src/v4_codec.cpp:26-40, src/v5_codec.cpp:719-740, :883-892); everything else is device plumbing: a
// grow-only workspace, the chunk table of a batch, kernel launches on one HIP stream.
//
// There is no CPU implementation behind this ABI: without a GPU every call fails with
// CLDN_HIP_ERR_NO_DEVICE / CLDN_HIP_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "cloudini_hip.h"
#include "stage1_device.h"
#include "stage1_launch.h"

using namespace cldn;

namespace {

thread_local std::string g_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess) {                                                                         \
      return fail(_e == hipErrorNoDevice ? CLDN_HIP_ERR_NO_DEVICE : CLDN_HIP_ERR_DEVICE,             \
                  "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);       \
    }                                                                                               \
  } while (0)

// Selects the codec's device for the duration of an ABI call and gives the calling thread its own device back
// (callers such as torch or a ROS node with several GPUs keep their own notion of the current device).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t enter(int device) {
    hipError_t e = hipGetDevice(&prev);
    if (e != hipSuccess) return e;
    if (prev == device) return hipSuccess;
    e = hipSetDevice(device);
    switched = (e == hipSuccess);
    return e;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};
#define ENTER_DEVICE(dev) \
  DeviceGuard _device_guard;  \
  HIP_TRY(_device_guard.enter(dev))

int size_of_type(uint8_t t) {  // include/cloudini_lib/basic_types.hpp:73-95
  switch (t) {
    case 1: case 2: return 1;
    case 3: case 4: return 2;
    case 5: case 6: case 7: return 4;
    case 8: case 9: case 10: return 8;
    default: return 0;
  }
}
bool is_adaptive_int(uint8_t t) {  // src/v5_codec.cpp:83-95
  return t == 3 || t == 4 || t == 5 || t == 6 || t == 9 || t == 10;
}

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return CLDN_HIP_OK;
    if (p) {
      HIP_TRY(hipFree(p));
      p = nullptr;
      cap = 0;
    }
    const size_t want = (bytes + 255) & ~size_t(255);
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      p = nullptr;
      return fail(e == hipErrorOutOfMemory ? CLDN_HIP_ERR_NOMEM : CLDN_HIP_ERR_DEVICE,
                  "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    cap = want;
    return CLDN_HIP_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct PinnedBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return CLDN_HIP_OK;
    if (p) {
      HIP_TRY(hipHostFree(p));
      p = nullptr;
      cap = 0;
    }
    const size_t want = (bytes + 4095) & ~size_t(4095);
    HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
    cap = want;
    return CLDN_HIP_OK;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

}  // namespace

namespace cldn {
int launch_fail(hipError_t e, const char* what) {
  return fail(e == hipErrorNoDevice ? CLDN_HIP_ERR_NO_DEVICE : CLDN_HIP_ERR_DEVICE, "%s failed: %s", what, hipGetErrorString(e));
}
}  // namespace cldn

struct cldn_hip_plan {
  DevPlan dev;
  std::vector<cldn_hip_field_t> fields;
  uint32_t point_step = 0;
  uint8_t version = 0;
  uint8_t encoding_opt = 0;
  bool uses_v5 = false;
  uint32_t ref_max_point_bytes = 0;  // detail::MaxSerializedPointSize
  bool has_padding = false;          // some byte of a point is not covered by any field
  // WIDE route (stage1_wide.h): the schema does not fit the launch-argument plan (more than kMaxOps regular tokens,
  // kMaxAdaptive adaptive fields or kMaxPointStep bytes per point). `dev` then keeps only its scalar members (n_ops,
  // n_adaptive and n_gorilla are 0: nothing on the host walks its arrays), the entries are here
  bool wide = false;
  std::vector<DevOp> ops_all;
  std::vector<uint32_t> op_aux;          // OP_GORILLA64: index of the op's token buffer
  std::vector<DevAdaptive> adaptive_all;
  uint32_t n_gorilla_all = 0;
  uint32_t n_adaptive_total() const { return wide ? (uint32_t)adaptive_all.size() : dev.n_adaptive; }
};

struct cldn_hip_codec {
  cldn_hip_plan plan;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // workspace (grow-only)
  DevBuf d_in, d_out, d_slots, d_chunks, d_cloud_first, d_payload, d_dst, d_offsets, d_modes;
  // zeroed once per encode call: [status block 256 B | k_finish anchors | section-handled flags | segment table]
  DevBuf d_status;
  DevBuf d_cols[kMaxAdaptive];
  DevBuf d_ranks[kMaxAdaptive];
  DevBuf d_dec_meta, d_pre_ptrs;
  DevBuf d_dec_cols[8];       // decode: dense columns of the adaptive fields that k_decode_points takes its integer fields from
  // stage 2 on the device (cldn_hip_codec_set_stage2): the stage-1 streams stay in d_s1, the LZ4 blocks go straight into the output
  // chunk table of the last cldn_hip_encode_stage1_chunks call (cldn_hip_frame_chunks frames it)
  bool ct_valid = false;
  uint32_t ct_n_chunks = 0, ct_n_clouds = 0, ct_segs_per_chunk = 0;
  uint64_t ct_slot_stride = 0, ct_need = 0;
  size_t ct_segs_off = 0, ct_anchor_off = 0;
  int stage2 = 0;
  DevBuf d_s1, d_s1_offsets, d_lz_matches, d_lz_counts, d_payload2, d_dst2, d_dec_split;
  DevBuf d_finrec;            // k_finish look-back records (rec, rec2), cleared only when (re)allocated
  int decode_fill = CLDN_HIP_FILL_KEEP;  // cldn_hip_codec_set_decode_fill
  DevBuf d_dec_bits;          // k_mark_token_ends: token-end bitmap of the streams of a decode call
  DevBuf d_dec_secs;          // k_section_offsets: per (field, chunk) section records + per-chunk counters
  DevBuf d_dec_rec;           // k_sections_cols_fast slice records, tagged with dec_epoch, cleared only when (re)allocated
  uint32_t dec_epoch = 0;
  uint32_t finish_epoch = 0;  // tag of this call's records
  bool force_ticket = false;      // k_finish waited in vain once (ST_FINISH_TIMEOUT): from then on its workgroups take tickets
  bool test_timeout_once = false; // cldn_hip_debug_finish_timeout_once: the next encode call's first attempt reports the timeout
  uint32_t test_split_parts = 0;  // cldn_hip_debug_decode_split: 0 = by the batch's size, 1 = chained launch, 2..16 = workgroups per chunk
  uint32_t finish_retries = 0;    // calls redone through the ticket path
  DevBuf d_pieces;  // piece table of the piece kernel (stage1_fused.h)
  uint32_t n_pieces = 0;
  uint32_t last_piece_pts = 0;
  bool last_quad_major = false;
  uint64_t pending_total = 0;  // bytes of a deferred host output waiting in d_out (cldn_hip_codec_fetch_output)
  int pipeline = 0;           // cldn_hip_codec_pipeline: 0 auto, 1 tile kernel + slots, 2 piece kernel + slots
  DevBuf d_viz_keys, d_viz_first, d_viz_slot, d_viz_blocks, d_viz_total;  // applyVizLossyPreprocessing workspace
  DevBuf d_pre[kMaxGorilla];
  // WIDE route: the plan's arrays in device memory (uploaded by cldn_hip_codec_create), per-chunk scratch of the encoder,
  // per-op state of the serial decoder, Gorilla token buffers
  DevBuf d_wide_plan, d_wide_scratch, d_wide_state;
  std::vector<DevBuf> d_wide_pre;
  WidePlan wide_desc = {};
  PinnedBuf h_stage;   // chunk table upload
  // decode table upload: a ring of staging buffers, each guarded by the event recorded behind its copy, so that a call
  // does not have to drain the stream before it fills the next one
  static constexpr int kDecStageRing = 4;
  PinnedBuf h_dec_stage[kDecStageRing];
  hipEvent_t dec_stage_ev[kDecStageRing] = {nullptr, nullptr, nullptr, nullptr};
  uint32_t dec_stage_next = 0;
  PinnedBuf h_result;  // offsets / status readback
  PinnedBuf h_modes;   // forced modes upload
  // modes of the previous encode call, copied back asynchronously: launch hint for the section kernels
  PinnedBuf h_last_modes;
  hipEvent_t ev_last_modes = nullptr;
  size_t last_modes_count = 0;  // n_clouds * n_adaptive of the copy in flight (0 = none)
  uint32_t last_modes_fields = 0;
  uint8_t hint_cache[kMaxAdaptive];  // modes seen by the most recent call whose copy has landed
  bool hint_valid = false;
  // cached batch shape
  std::vector<uint64_t> last_cloud_points;
  uint32_t last_n_chunks = 0;
  // timing
  std::vector<hipEvent_t> events;  // 5 per timing slot
  // decode: the route counters of an earlier call, copied back asynchronously (like the encoder's mode hint): when every chunk's
  // section was a small Palette the point kernel found by itself, the next call does not launch the section pre-kernels
  PinnedBuf h_dec_stats;
  hipEvent_t ev_dec_stats = nullptr;
  uint32_t dec_stats_chunks = 0;     // chunks of the call whose copy is in flight (0 = none)
  bool dec_palette_hint = false;
  bool dec_stats_seen = false;       // one copy of the counters has landed
  uint32_t dec_dv_hint = 0;          // 1: the last counters showed no DeltaVarint section decoded by k_section_dv_w, 2: every chunk's (stage1_launch.h)
  uint32_t dec_call_index = 0;
  hipEvent_t dec_events[4] = {nullptr, nullptr, nullptr, nullptr};  // the last decode call's (timing enabled)
  bool dec_events_valid = false;
  std::vector<uint8_t> slot_valid;
  uint64_t call_index = 0;
  // modes committed elsewhere (continuation of a cloud from a chunk boundary); empty = probe
  std::vector<uint8_t> forced_modes;
};

extern "C" {

const char* cldn_hip_last_error(void) { return g_error.c_str(); }
int cldn_hip_abi_version(void) { return CLDN_HIP_ABI_VERSION; }

int cldn_hip_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e == hipErrorNoDevice) return 0;
  if (e != hipSuccess) return fail(CLDN_HIP_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

void* cldn_hip_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) {  // page-locked for every device
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
void cldn_hip_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int cldn_hip_current_device(void) {
  int d = -1;
  hipError_t e = hipGetDevice(&d);
  if (e != hipSuccess)
    return fail(e == hipErrorNoDevice ? CLDN_HIP_ERR_NO_DEVICE : CLDN_HIP_ERR_DEVICE, "hipGetDevice: %s", hipGetErrorString(e));
  return d;
}

int cldn_hip_set_current_device(int device) {
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess)
    return fail(e == hipErrorNoDevice ? CLDN_HIP_ERR_NO_DEVICE : CLDN_HIP_ERR_DEVICE, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
  return CLDN_HIP_OK;
}

int cldn_hip_plan_create(const cldn_hip_field_t* fields, uint32_t n_fields, uint32_t point_step, uint8_t version,
                         uint8_t encoding_opt, cldn_hip_plan_t** out) {
  if (!out) return fail(CLDN_HIP_ERR_ARG, "plan_create: out is NULL");
  *out = nullptr;
  if (point_step == 0) return fail(CLDN_HIP_ERR_ARG, "point_step cannot be 0");  // cloudini.cpp:250-252
  if (n_fields && !fields) return fail(CLDN_HIP_ERR_ARG, "fields is NULL");
  if (encoding_opt > 2) return fail(CLDN_HIP_ERR_ARG, "invalid encoding_opt %u", encoding_opt);

  cldn_hip_plan* plan = new (std::nothrow) cldn_hip_plan();
  if (!plan) return fail(CLDN_HIP_ERR_NOMEM, "out of memory");
  plan->fields.assign(fields, fields + n_fields);
  plan->point_step = point_step;
  plan->version = version;
  plan->encoding_opt = encoding_opt;
  DevPlan& d = plan->dev;
  memset(&d, 0, sizeof(d));
  d.point_step = point_step;
  const bool lossy = encoding_opt == 1;

  // MaxSerializedPointSize (codec_common.cpp:29-67) -- also validates the field types
  for (uint32_t i = 0; i < n_fields; ++i) {
    const cldn_hip_field_t& f = fields[i];
    const int sz = size_of_type(f.type);
    if (sz == 0) {
      delete plan;
      return fail(CLDN_HIP_ERR_ARG, "Unsupported field type %u (field %u)", f.type, i);
    }
    if ((uint64_t)f.offset + (uint64_t)sz > point_step) {
      delete plan;
      return fail(CLDN_HIP_ERR_ARG, "field %u (offset %u, %d bytes) exceeds point_step %u", i, f.offset, sz,
                  point_step);
    }
    switch (f.type) {
      case 7: plan->ref_max_point_bytes += (lossy && f.has_resolution) ? 10 : 7; break;
      case 8: plan->ref_max_point_bytes += (lossy && f.has_resolution) ? 10 : 11; break;
      case 1: case 2: plan->ref_max_point_bytes += 1; break;
      default: plan->ref_max_point_bytes += 10; break;
    }
  }

  {
    std::vector<uint8_t> covered(point_step, 0);
    for (uint32_t i = 0; i < n_fields; ++i)
      for (int b = 0; b < size_of_type(fields[i].type); ++b) covered[fields[i].offset + b] = 1;
    plan->has_padding = std::find(covered.begin(), covered.end(), 0) != covered.end();
  }
  // LeadingLossyFloatFieldCount (codec_common.cpp:69-82)
  uint32_t lead = 0;
  if (lossy) {
    for (uint32_t i = 0; i < n_fields; ++i) {
      if (fields[i].type != 7 || !fields[i].has_resolution) break;
      ++lead;
    }
    if (lead != 3 && lead != 4) lead = 0;
  }
  // UsesV5Codec (v5_codec.cpp:883-892)
  plan->uses_v5 = false;
  if (version >= 5 && lossy) {
    for (uint32_t i = lead; i < n_fields; ++i) plan->uses_v5 |= is_adaptive_int(fields[i].type);
  }

  // (every op and adaptive field goes to the plan's vectors first; they move into `d`'s arrays at the end if they fit)
  auto add_op = [&](DevOp op) -> int {
    plan->ops_all.push_back(op);
    plan->op_aux.push_back(op.kind == OP_GORILLA64 ? plan->n_gorilla_all++ : 0u);
    d.max_regular_bytes += op.max_bytes;
    return CLDN_HIP_OK;
  };
  int rc = CLDN_HIP_OK;
  d.all_varint = 1;
  for (uint32_t i = 0; i < n_fields && rc == CLDN_HIP_OK; ++i) {
    const cldn_hip_field_t& f = fields[i];
    DevOp op;
    memset(&op, 0, sizeof(op));
    op.offset = f.offset;
    op.type = f.type;
    op.size = (uint8_t)size_of_type(f.type);
    if (encoding_opt == 0) {  // BuildV4Encoders, v4_codec.cpp:29-34: everything is a raw copy
      op.kind = OP_COPY;
      op.max_bytes = op.size;
      d.all_varint = 0;
      d.min_regular_bytes += op.size;
      rc = add_op(op);
      continue;
    }
    if (i < lead) {  // FieldEncoderFloatN_Lossy lane, field_encoder.cpp:24-40
      op.kind = OP_QF32;
      op.mult_f = 1.0F / f.resolution;
      op.res_f = f.resolution;
      op.max_bytes = 5;
      if (!(op.mult_f > 0.0f)) {
        rc = fail(CLDN_HIP_ERR_ARG, "FieldEncoderFloatN_Lossy requires a resolution with value > 0.0");
        break;
      }
      d.min_regular_bytes += 1;
      rc = add_op(op);
      continue;
    }
    if (plan->uses_v5 && is_adaptive_int(f.type)) {  // buildV5Plan, v5_codec.cpp:725-737
      DevAdaptive a;
      memset(&a, 0, sizeof(a));
      a.offset = f.offset;
      a.type = f.type;
      a.bpv = op.size;
      plan->adaptive_all.push_back(a);
      continue;
    }
    switch (f.type) {  // CreateCompatibleEncoder, codec_common.cpp:116-153
      case 7:
        if (lossy && f.has_resolution) {
          if (!(f.resolution > 0.0f)) {
            rc = fail(CLDN_HIP_ERR_ARG, "FieldEncoder(Float/Lossy) requires a resolution with value > 0.0");
            break;
          }
          op.kind = OP_LOSSY_F32;
          op.mult_f = (float)(1.0 / (double)f.resolution);  // field_encoder.hpp:101-102
          op.res_f = f.resolution;
          op.max_bytes = 10;
          d.min_regular_bytes += 1;
        } else if (encoding_opt == 2) {
          op.kind = OP_XOR32;
          op.max_bytes = 4;
          d.all_varint = 0;
          d.min_regular_bytes += 4;
        } else {
          op.kind = OP_COPY;
          op.max_bytes = 4;
          d.all_varint = 0;
          d.min_regular_bytes += 4;
        }
        break;
      case 8:
        if (lossy && f.has_resolution) {
          if (!(f.resolution > 0.0f)) {
            rc = fail(CLDN_HIP_ERR_ARG, "FieldEncoder(Float/Lossy) requires a resolution with value > 0.0");
            break;
          }
          op.kind = OP_LOSSY_F64;
          op.mult_d = 1.0 / (double)f.resolution;
          op.res_d = (double)f.resolution;
          op.max_bytes = 10;
          d.min_regular_bytes += 1;
        } else if (!f.has_resolution && version >= 4) {
          op.kind = OP_GORILLA64;  // FieldEncoderFloat_Gorilla<double>
          op.type = (uint8_t)(plan->n_gorilla_all & 0xffu);  // index of the op's token buffer (the WIDE route reads op_aux instead)
          op.max_bytes = 10;       // 1 + 1 + 5 + 6 + 64 = 77 bits
          d.all_varint = 0;
          d.min_regular_bytes += 1;
        } else {
          op.kind = OP_XOR64;
          op.max_bytes = 8;
          d.all_varint = 0;
          d.min_regular_bytes += 8;
        }
        break;
      case 1: case 2:
        op.kind = OP_COPY;
        op.max_bytes = 1;
        d.all_varint = 0;
        d.min_regular_bytes += 1;
        break;
      default:
        op.kind = OP_INT;
        op.max_bytes = 10;
        d.min_regular_bytes += 1;
        break;
    }
    if (rc == CLDN_HIP_OK) rc = add_op(op);
  }
  if (rc != CLDN_HIP_OK) {
    delete plan;
    return rc;
  }
  plan->wide = plan->ops_all.size() > (size_t)kMaxOps || plan->adaptive_all.size() > (size_t)kMaxAdaptive ||
               plan->n_gorilla_all > (uint32_t)kMaxGorilla || point_step > kMaxPointStep;
  if (!plan->wide) {
    d.n_ops = (uint32_t)plan->ops_all.size();
    for (uint32_t k = 0; k < d.n_ops; ++k) d.ops[k] = plan->ops_all[k];
    d.n_adaptive = (uint32_t)plan->adaptive_all.size();
    for (uint32_t a = 0; a < d.n_adaptive; ++a) d.adaptive[a] = plan->adaptive_all[a];
    d.n_gorilla = plan->n_gorilla_all;
  } else {
    d.all_varint = 0;  // (no kernel of the ordinary route may take this plan)
  }
  {  // decode: can the token ends of the regular stream be laid out from the point's form (k_mark_token_ends)?
    bool ok = d.n_ops >= 1u && d.n_ops <= 8u && d.max_regular_bytes <= 256u;
    uint32_t n_raw = 0u;
    for (uint32_t o = 0; o < d.n_ops && ok; ++o) {
      const uint32_t k = d.ops[o].kind;
      if (k == OP_COPY || k == OP_XOR32 || k == OP_XOR64) {
        const uint32_t sz = d.ops[o].size;
        ok = sz == 1u || sz == 2u || sz == 4u || sz == 8u;
        ++n_raw;
      } else {
        ok = k == OP_QF32 || k == OP_LOSSY_F32 || k == OP_LOSSY_F64 || k == OP_INT;
      }
    }
    d.varint_and_raw = (ok && n_raw != 0u) ? 1u : 0u;
  }
  *out = plan;
  return CLDN_HIP_OK;
}

void cldn_hip_plan_destroy(cldn_hip_plan_t* plan) { delete plan; }
int cldn_hip_plan_uses_v5(const cldn_hip_plan_t* plan) { return plan && plan->uses_v5 ? 1 : 0; }
uint32_t cldn_hip_plan_adaptive_fields(const cldn_hip_plan_t* plan) { return plan ? plan->n_adaptive_total() : 0; }
uint32_t cldn_hip_plan_max_point_bytes(const cldn_hip_plan_t* plan) { return plan ? plan->ref_max_point_bytes : 0; }

uint64_t cldn_hip_stage1_bound(const cldn_hip_plan_t* plan, uint64_t n_points) {  // cloudini.cpp:249-292 (NONE)
  if (!plan) return 0;
  uint64_t total = 0, left = n_points;
  while (left > 0) {
    const uint64_t in_chunk = std::min<uint64_t>(left, kPointsPerChunk);
    left -= in_chunk;
    uint64_t chunk = in_chunk * plan->ref_max_point_bytes;
    if (plan->uses_v5) chunk += (uint64_t)plan->fields.size() * 32u + 1024u;
    total += 4 + chunk;
  }
  return total;
}

static uint64_t lz4_block_bound(uint64_t n) { return n + n / 255u + 16u; }  // LZ4_COMPRESSBOUND, lz4.h

uint64_t cldn_hip_stage2_bound(const cldn_hip_plan_t* plan, uint64_t n_points, int stage2) {  // cloudini.cpp:249-292
  if (!plan) return 0;
  if (stage2 == CLDN_HIP_STAGE2_NONE) return cldn_hip_stage1_bound(plan, n_points);
  uint64_t total = 0, left = n_points;
  while (left > 0) {
    const uint64_t in_chunk = std::min<uint64_t>(left, kPointsPerChunk);
    left -= in_chunk;
    uint64_t chunk = in_chunk * plan->ref_max_point_bytes;
    if (plan->uses_v5) chunk += (uint64_t)plan->fields.size() * 32u + 1024u;
    total += 4 + lz4_block_bound(chunk);
  }
  return total;
}

int cldn_hip_codec_set_stage2(cldn_hip_codec_t* c, int stage2) {
  if (!c) return fail(CLDN_HIP_ERR_ARG, "codec is NULL");
  if (stage2 != CLDN_HIP_STAGE2_NONE && stage2 != CLDN_HIP_STAGE2_LZ4 && stage2 != CLDN_HIP_STAGE2_LZ4_FAST)
    return fail(CLDN_HIP_ERR_ARG, "invalid stage-2 mode %d", stage2);
  c->stage2 = stage2;
  return CLDN_HIP_OK;
}

int cldn_hip_codec_create(const cldn_hip_plan_t* plan, int device, void* hip_stream, cldn_hip_codec_t** out) {
  if (!plan || !out) return fail(CLDN_HIP_ERR_ARG, "codec_create: NULL argument");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(CLDN_HIP_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                hipGetErrorString(e));
  if (device < 0) HIP_TRY(hipGetDevice(&device));
  if (device >= n) return fail(CLDN_HIP_ERR_ARG, "device %d out of range (%d devices)", device, n);
  ENTER_DEVICE(device);
  cldn_hip_codec* c = new (std::nothrow) cldn_hip_codec();
  if (!c) return fail(CLDN_HIP_ERR_NOMEM, "out of memory");
  c->plan = *plan;
  c->device = device;
  if (hip_stream) {
    c->stream = (hipStream_t)hip_stream;
  } else {
    hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
      delete c;
      return fail(CLDN_HIP_ERR_DEVICE, "hipStreamCreate: %s", hipGetErrorString(se));
    }
    c->own_stream = true;
  }
  int rc = stage1_configure_kernels();
  if (rc == CLDN_HIP_OK && c->plan.wide) {
    // WIDE route: [ops | op_aux | adaptive] in one device buffer, the descriptor that points into it
    const cldn_hip_plan& P = c->plan;
    const size_t ops_b = (P.ops_all.size() * sizeof(DevOp) + 63u) & ~size_t(63);
    const size_t aux_b = (P.op_aux.size() * sizeof(uint32_t) + 63u) & ~size_t(63);
    const size_t ada_b = (P.adaptive_all.size() * sizeof(DevAdaptive) + 63u) & ~size_t(63);
    rc = c->d_wide_plan.ensure(ops_b + aux_b + ada_b + 64u);
    if (rc == CLDN_HIP_OK) {
      uint8_t* base = (uint8_t*)c->d_wide_plan.p;
      hipError_t he = hipSuccess;
      if (!P.ops_all.empty()) he = hipMemcpy(base, P.ops_all.data(), P.ops_all.size() * sizeof(DevOp), hipMemcpyHostToDevice);
      if (he == hipSuccess && !P.op_aux.empty())
        he = hipMemcpy(base + ops_b, P.op_aux.data(), P.op_aux.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
      if (he == hipSuccess && !P.adaptive_all.empty())
        he = hipMemcpy(base + ops_b + aux_b, P.adaptive_all.data(), P.adaptive_all.size() * sizeof(DevAdaptive), hipMemcpyHostToDevice);
      if (he != hipSuccess) rc = fail(CLDN_HIP_ERR_DEVICE, "upload of the wide plan: %s", hipGetErrorString(he));
      WidePlan& W = c->wide_desc;
      W.point_step = P.point_step;
      W.n_ops = (uint32_t)P.ops_all.size();
      W.n_adaptive = (uint32_t)P.adaptive_all.size();
      W.n_gorilla = P.n_gorilla_all;
      W.min_regular_bytes = P.dev.min_regular_bytes;
      W.reserved = 0u;
      W.ops = reinterpret_cast<const DevOp*>(base);
      W.op_aux = reinterpret_cast<const uint32_t*>(base + ops_b);
      W.adaptive = reinterpret_cast<const DevAdaptive*>(base + ops_b + aux_b);
      c->d_wide_pre.resize(P.n_gorilla_all);
    }
  }
  if (rc != CLDN_HIP_OK) {
    cldn_hip_codec_destroy(c);
    return rc;
  }
  *out = c;
  return CLDN_HIP_OK;
}

void cldn_hip_codec_destroy(cldn_hip_codec_t* c) {
  if (!c) return;
  DeviceGuard guard;
  (void)guard.enter(c->device);
  (void)hipStreamSynchronize(c->stream);
  DevBuf* bufs[] = {&c->d_in, &c->d_out, &c->d_slots, &c->d_chunks, &c->d_cloud_first, &c->d_finrec, &c->d_dec_rec, &c->d_dec_bits, &c->d_dec_secs, &c->d_s1, &c->d_s1_offsets,
                    &c->d_lz_matches, &c->d_lz_counts, &c->d_payload2, &c->d_dst2, &c->d_dec_split,
                    &c->d_payload, &c->d_dst, &c->d_offsets, &c->d_modes, &c->d_status, &c->d_dec_meta, &c->d_pre_ptrs, &c->d_dec_cols[0], &c->d_dec_cols[1], &c->d_dec_cols[2], &c->d_dec_cols[3], &c->d_dec_cols[4], &c->d_dec_cols[5], &c->d_dec_cols[6], &c->d_dec_cols[7],
                    &c->d_viz_keys, &c->d_viz_first,
                    &c->d_viz_slot, &c->d_viz_blocks, &c->d_viz_total, &c->d_pieces};
  for (DevBuf* b : bufs) b->release();
  for (int a = 0; a < kMaxAdaptive; ++a) {
    c->d_cols[a].release();
    c->d_ranks[a].release();
  }
  for (int g = 0; g < kMaxGorilla; ++g) c->d_pre[g].release();
  c->d_wide_plan.release();
  c->d_wide_scratch.release();
  c->d_wide_state.release();
  for (DevBuf& b : c->d_wide_pre) b.release();
  c->h_stage.release();
  for (int k = 0; k < cldn_hip_codec::kDecStageRing; ++k) {
    c->h_dec_stage[k].release();
    if (c->dec_stage_ev[k]) (void)hipEventDestroy(c->dec_stage_ev[k]);
  }
  c->h_result.release();
  c->h_modes.release();
  c->h_last_modes.release();
  if (c->ev_last_modes) (void)hipEventDestroy(c->ev_last_modes);
  c->h_dec_stats.release();
  if (c->ev_dec_stats) (void)hipEventDestroy(c->ev_dec_stats);
  for (hipEvent_t& ev : c->events)
    if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t& ev : c->dec_events)
    if (ev) (void)hipEventDestroy(ev);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int cldn_hip_codec_synchronize(cldn_hip_codec_t* c) {
  if (!c) return fail(CLDN_HIP_ERR_ARG, "codec is NULL");
  HIP_TRY(hipStreamSynchronize(c->stream));
  return CLDN_HIP_OK;
}

void* cldn_hip_codec_stream(cldn_hip_codec_t* c) { return c ? (void*)c->stream : nullptr; }
int cldn_hip_codec_device(const cldn_hip_codec_t* c) { return c ? c->device : CLDN_HIP_ERR_ARG; }

int cldn_hip_codec_enable_timing(cldn_hip_codec_t* c, uint32_t n_slots) {
  if (!c) return fail(CLDN_HIP_ERR_ARG, "codec is NULL");
  ENTER_DEVICE(c->device);
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (hipEvent_t& ev : c->events)
    if (ev) (void)hipEventDestroy(ev);
  c->events.assign((size_t)n_slots * 5, nullptr);
  c->slot_valid.assign(n_slots, 0);
  for (hipEvent_t& ev : c->events) HIP_TRY(hipEventCreate(&ev));
  c->call_index = 0;
  c->dec_events_valid = false;
  for (hipEvent_t& ev : c->dec_events) {
    if (n_slots && !ev) HIP_TRY(hipEventCreate(&ev));
    if (!n_slots && ev) {
      (void)hipEventDestroy(ev);
      ev = nullptr;
    }
  }
  return CLDN_HIP_OK;
}

// Test hook, outside the boundary of include/cloudini_hip.h (like cldn_hip_debug_finish_trace): the codec's next encode call
// reports ST_FINISH_TIMEOUT on its first attempt, so that the retry through the ticket order can be exercised
// (tests/test_gpu_encode.py). Rounds 3-4 read an environment variable in the encode path for this.
__attribute__((visibility("default"))) int cldn_hip_debug_finish_timeout_once(cldn_hip_codec_t* c) {
  if (!c) return fail(CLDN_HIP_ERR_ARG, "codec is NULL");
  c->test_timeout_once = true;
  return CLDN_HIP_OK;
}

// Test hook, outside the boundary like the one above: which launch shape the point decoder takes (tests/test_gpu_decode.py
// runs every schema family through the chained and the SPLIT launches whatever the batch's size). parts: 0 = decided by the
// number of chunks (the default), 1 = the chained launch, 2..16 = a SPLIT launch with that many workgroups per chunk. Bytes
// never depend on it. (Rounds 4-5 read environment variables in the decode path for this.)
__attribute__((visibility("default"))) int cldn_hip_debug_decode_split(cldn_hip_codec_t* c, uint32_t parts) {
  if (!c) return fail(CLDN_HIP_ERR_ARG, "codec is NULL");
  if (parts > 16u) return fail(CLDN_HIP_ERR_ARG, "at most 16 workgroups per chunk");
  c->test_split_parts = parts;
  return CLDN_HIP_OK;
}

int cldn_hip_codec_decode_ms(cldn_hip_codec_t* c, float ms[2]) {
  if (!c || !ms) return fail(CLDN_HIP_ERR_ARG, "NULL argument");
  if (!c->dec_events_valid) return fail(CLDN_HIP_ERR_ARG, "no timed decode call (cldn_hip_codec_enable_timing first)");
  HIP_TRY(hipEventSynchronize(c->dec_events[3]));
  HIP_TRY(hipEventElapsedTime(&ms[0], c->dec_events[1], c->dec_events[2]));
  HIP_TRY(hipEventElapsedTime(&ms[1], c->dec_events[0], c->dec_events[3]));
  return CLDN_HIP_OK;
}

int cldn_hip_codec_kernel_ms(cldn_hip_codec_t* c, uint32_t slot, float ms[4]) {
  if (!c || !ms) return fail(CLDN_HIP_ERR_ARG, "NULL argument");
  if (slot >= c->slot_valid.size() || !c->slot_valid[slot]) return fail(CLDN_HIP_ERR_ARG, "no timed call in slot %u", slot);
  hipEvent_t* ev = &c->events[(size_t)slot * 5];
  HIP_TRY(hipEventSynchronize(ev[4]));
  HIP_TRY(hipEventElapsedTime(&ms[3], ev[0], ev[4]));
  HIP_TRY(hipEventElapsedTime(&ms[0], ev[1], ev[2]));
  HIP_TRY(hipEventElapsedTime(&ms[1], ev[2], ev[3]));
  HIP_TRY(hipEventElapsedTime(&ms[2], ev[3], ev[4]));
  return CLDN_HIP_OK;
}

int cldn_hip_codec_status(cldn_hip_codec_t* c) {
  if (!c) return fail(CLDN_HIP_ERR_ARG, "codec is NULL");
  ENTER_DEVICE(c->device);
  if (!c->d_status.p) return CLDN_HIP_OK;
  uint32_t st = 0;
  HIP_TRY(hipMemcpyAsync(&st, c->d_status.p, sizeof(st), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (st & ST_FINISH_TIMEOUT) {
    c->force_ticket = true;  // (device-resident outputs cannot be redone here: the caller repeats the call, which then takes tickets)
    return fail(CLDN_HIP_ERR_DEVICE, "k_finish: a workgroup waited too long for the sizes of the chunks before it (status 0x%x); the codec uses the ticket order from now on, repeat the call", st);
  }
  if (st & ST_OUT_OVERFLOW) return fail(CLDN_HIP_ERR_CAPACITY, "Output buffer too small for the encoded stream");
  if (st & ST_CORRUPT) return fail(CLDN_HIP_ERR_CORRUPT, "malformed stage-1 stream");
  return CLDN_HIP_OK;
}

// Build (or reuse) the chunk table of a batch. Returns the number of chunks and total points.
// piece_pts != 0: also the piece table of the single-pass encoder (pieces of piece_pts points, per chunk padded to 4)
// quad_major: the piece table lists workgroup 0 (pieces 0..3) of every chunk, then workgroup 1 of every chunk, ... so
// that the workgroups of ONE chunk start far apart in time (intra-chunk placement of the piece kernel)
static int upload_batch_shape(cldn_hip_codec* c, const uint64_t* cloud_points, uint32_t n_clouds, uint32_t piece_pts,
                              bool quad_major, uint32_t* n_chunks_out, uint64_t* n_points_out) {
  uint64_t total_points = 0;
  uint64_t n_chunks64 = 0;
  for (uint32_t k = 0; k < n_clouds; ++k) {
    total_points += cloud_points[k];
    n_chunks64 += (cloud_points[k] + kPointsPerChunk - 1) / kPointsPerChunk;
  }
  if (n_chunks64 > 0x3fffffffull) return fail(CLDN_HIP_ERR_ARG, "batch too large");
  const uint32_t n_chunks = (uint32_t)n_chunks64;
  *n_chunks_out = n_chunks;
  *n_points_out = total_points;
  int rc;
  const bool same = c->last_cloud_points.size() == n_clouds && c->last_n_chunks == n_chunks &&
                    c->last_piece_pts == piece_pts && c->last_quad_major == quad_major &&
                    std::equal(c->last_cloud_points.begin(), c->last_cloud_points.end(), cloud_points);
  uint64_t n_pieces64 = 0;
  if (piece_pts) {
    for (uint32_t k = 0; k < n_clouds; ++k) {
      const uint64_t full = cloud_points[k] / kPointsPerChunk, rest = cloud_points[k] % kPointsPerChunk;
      n_pieces64 += full * ((((uint64_t)kPointsPerChunk + piece_pts - 1) / piece_pts + 3) & ~uint64_t(3));
      if (rest) n_pieces64 += ((rest + piece_pts - 1) / piece_pts + 3) & ~uint64_t(3);
    }
    if (n_pieces64 > 0x7fffffffull) return fail(CLDN_HIP_ERR_ARG, "batch too large");
  }
  c->n_pieces = (uint32_t)n_pieces64;
  if (piece_pts && (rc = c->d_pieces.ensure(std::max<size_t>(1, (size_t)n_pieces64) * sizeof(PieceDesc))) != CLDN_HIP_OK) return rc;
  if ((rc = c->d_chunks.ensure(std::max<size_t>(1, n_chunks) * sizeof(ChunkDesc))) != CLDN_HIP_OK) return rc;
  if ((rc = c->d_cloud_first.ensure((size_t)(n_clouds + 1) * sizeof(uint32_t))) != CLDN_HIP_OK) return rc;
  if (same) return CLDN_HIP_OK;

  const size_t head_bytes = ((size_t)n_chunks * sizeof(ChunkDesc) + (size_t)(n_clouds + 1) * sizeof(uint32_t) + 31) & ~size_t(31);
  const size_t bytes = head_bytes + (size_t)n_pieces64 * sizeof(PieceDesc);
  // the staging buffer may still be in flight from the previous (asynchronous) call
  HIP_TRY(hipStreamSynchronize(c->stream));
  if ((rc = c->h_stage.ensure(bytes)) != CLDN_HIP_OK) return rc;
  ChunkDesc* hc = (ChunkDesc*)c->h_stage.p;
  uint32_t* hf = (uint32_t*)((uint8_t*)c->h_stage.p + (size_t)n_chunks * sizeof(ChunkDesc));
  uint64_t first = 0;
  uint32_t ci = 0;
  for (uint32_t k = 0; k < n_clouds; ++k) {
    hf[k] = ci;
    uint64_t left = cloud_points[k];
    uint32_t j = 0;
    while (left > 0) {
      const uint32_t n = (uint32_t)std::min<uint64_t>(left, kPointsPerChunk);
      ChunkDesc& cd = hc[ci++];
      cd.first_point = first;
      cd.n_points = n;
      cd.cloud = k;
      cd.chunk_in_cloud = j++;
      cd.reserved = 0;
      first += n;
      left -= n;
    }
  }
  hf[n_clouds] = ci;
  if (piece_pts && n_pieces64) {
    PieceDesc* hp = (PieceDesc*)((uint8_t*)c->h_stage.p + head_bytes);
    size_t g = 0;
    auto put = [&](uint32_t k, uint32_t q, uint32_t P) {
      hp[g].chunk_first_point = hc[k].first_point;
      hp[g].chunk = k;
      hp[g].cloud = hc[k].cloud;
      hp[g].n_chunk_points = hc[k].n_points;
      hp[g].p = (uint16_t)q;
      hp[g].P = (uint16_t)P;
      hp[g].pad[0] = hp[g].pad[1] = 0;
      ++g;
    };
    const uint32_t max_p = (((kPointsPerChunk + piece_pts - 1) / piece_pts) + 3u) & ~3u;
    if (!quad_major) {
      for (uint32_t k = 0; k < n_chunks; ++k) {
        const uint32_t P = (((hc[k].n_points + piece_pts - 1) / piece_pts) + 3u) & ~3u;
        for (uint32_t q = 0; q < P; ++q) put(k, q, P);
      }
    } else {
      for (uint32_t q0 = 0; q0 < max_p; q0 += 4u)
        for (uint32_t k = 0; k < n_chunks; ++k) {
          const uint32_t P = (((hc[k].n_points + piece_pts - 1) / piece_pts) + 3u) & ~3u;
          if (q0 < P)
            for (uint32_t q = q0; q < q0 + 4u; ++q) put(k, q, P);
        }
    }
    HIP_TRY(hipMemcpyAsync(c->d_pieces.p, hp, g * sizeof(PieceDesc), hipMemcpyHostToDevice, c->stream));
  }
  c->last_piece_pts = piece_pts;
  c->last_quad_major = quad_major;
  if (n_chunks)
    HIP_TRY(hipMemcpyAsync(c->d_chunks.p, hc, (size_t)n_chunks * sizeof(ChunkDesc), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_cloud_first.p, hf, (size_t)(n_clouds + 1) * sizeof(uint32_t), hipMemcpyHostToDevice,
                         c->stream));
  c->last_cloud_points.assign(cloud_points, cloud_points + n_clouds);
  c->last_n_chunks = n_chunks;
  return CLDN_HIP_OK;
}

}  // extern "C"

// cloud_ptrs != NULL: the clouds live in separate HOST buffers (`points` is ignored)
// table != NULL: chunk-table output (cldn_hip_encode_stage1_chunks): no framing, `out` is not used
constexpr int kRetryWithTicket = 0x7fff0001;  // encode_stage1_once: k_finish timed out without the ticket, the call is redone with it

static int encode_stage1_once(cldn_hip_codec_t* c, const void* points, int points_loc, const void* const* cloud_ptrs,
                              const uint64_t* cloud_points, uint32_t n_clouds, void* out, uint64_t out_capacity, int out_loc,
                              uint64_t* stream_offsets, uint32_t* chunk_sizes, uint8_t* modes,
                              cldn_hip_chunk_table_t* table = nullptr) {
  if (!c) return fail(CLDN_HIP_ERR_ARG, "codec is NULL");
  if (table) {
    memset(table, 0, sizeof(*table));
    c->ct_valid = false;
    if (c->stage2 != CLDN_HIP_STAGE2_NONE) return fail(CLDN_HIP_ERR_ARG, "chunk-table output and stage 2 on the device exclude each other");
  }
  if (n_clouds && !cloud_points) return fail(CLDN_HIP_ERR_ARG, "cloud_points is NULL");
  if ((points_loc != CLDN_HIP_HOST && points_loc != CLDN_HIP_DEVICE) ||
      (out_loc != CLDN_HIP_HOST && out_loc != CLDN_HIP_DEVICE))
    return fail(CLDN_HIP_ERR_ARG, "invalid memory location tag");
  ENTER_DEVICE(c->device);
  c->ct_valid = false;  // (the workspace is about to be rewritten)
  const DevPlan& plan = c->plan.dev;
  const uint32_t step = plan.point_step;

  uint32_t n_chunks = 0;
  uint64_t n_points = 0;
  // device-resident inputs decide the kernel variant by their address; host inputs are staged into an aligned buffer
  const uint8_t* variant_ptr = points_loc == CLDN_HIP_DEVICE ? (const uint8_t*)points : nullptr;
  const bool wide = c->plan.wide;  // stage1_wide.h: the plan's arrays live in device memory, one segment per chunk
  const uint32_t piece_pts = (c->pipeline == 1 || wide) ? 0u : stage1_piece_points(plan, variant_ptr);
  static const bool intra_env0 = dev_env_int("CLDN_HIP_INTRA", 0) != 0;  // A/B switch
  const bool intra_env = intra_env0 || table != nullptr;  // chunk tables want one regular segment per chunk
  static const bool quad_major_env = dev_env_int("CLDN_HIP_QUAD_MAJOR", 1) != 0;
  int rc = upload_batch_shape(c, cloud_points, n_clouds, piece_pts, piece_pts != 0u && intra_env && quad_major_env, &n_chunks, &n_points);
  if (rc != CLDN_HIP_OK) return rc;
  if (n_points && !points && !cloud_ptrs) return fail(CLDN_HIP_ERR_ARG, "points is NULL");
  if (cloud_ptrs)
    for (uint32_t k = 0; k < n_clouds; ++k)
      if (cloud_points[k] && !cloud_ptrs[k]) return fail(CLDN_HIP_ERR_ARG, "cloud %u: NULL buffer", k);

  // capacity contract of PointcloudEncoder::encode (cloudini.cpp:531-534)
  uint64_t need = 0, need_s1 = 0;
  for (uint32_t k = 0; k < n_clouds; ++k) {
    need_s1 += cldn_hip_stage1_bound(&c->plan, cloud_points[k]);
    need += cldn_hip_stage2_bound(&c->plan, cloud_points[k], c->stage2);
  }
  const bool lz4 = c->stage2 == CLDN_HIP_STAGE2_LZ4 || c->stage2 == CLDN_HIP_STAGE2_LZ4_FAST;
  const bool lz4_fast = c->stage2 == CLDN_HIP_STAGE2_LZ4_FAST;
  // two-step host output (cldn_hip_codec_fetch_output): no caller buffer yet, the codec's own device buffer takes the bound
  const bool deferred = out == nullptr && out_loc == CLDN_HIP_HOST && !table;
  if (deferred || table) out_capacity = need;
  if (out_capacity < need)
    return fail(CLDN_HIP_ERR_CAPACITY, "Output buffer too small for worst-case compressed size (%llu < %llu)",
                (unsigned long long)out_capacity, (unsigned long long)need);
  if (need && !out && !deferred && !table) return fail(CLDN_HIP_ERR_ARG, "out is NULL");
  c->pending_total = 0;

  const uint32_t n_adaptive = c->plan.n_adaptive_total();
  // Sub-chunks: the regular stream of a chunk is produced as `subs` independent sub-streams (one workgroup each)
  // that the compaction kernel concatenates; this multiplies the parallelism of small batches at no extra work.
  uint32_t subs = 1;
  if (const int forced = dev_env_int("CLDN_HIP_SUBCHUNKS", 0)) {
    subs = (uint32_t)forced;
  } else {
    while (subs < 32u && (uint64_t)n_chunks * subs < 6000u) subs *= 2u;
  }
  if (subs < 1u || subs > 32u || (subs & (subs - 1u)) || wide) subs = 1u;
  const bool pieces = piece_pts != 0u && n_chunks != 0u;   // regular stream by the piece kernel
  const bool intra = pieces && intra_env;
  const uint32_t piece_wgs = ((((kPointsPerChunk + piece_pts - 1u) / std::max(1u, piece_pts)) + 3u) & ~3u) / 4u;  // workgroups (4 pieces) per full chunk
  if (pieces) subs = intra ? 1u : piece_wgs;  // one segment per workgroup, or one per chunk
  const uint32_t sub_points = pieces ? piece_pts : kPointsPerChunk / subs;
  // (intra: the slot still reserves the worst case of every workgroup, the streams are packed at its start)
  const uint32_t sub_stride = pieces ? 4u * stage1_piece_slot_stride(plan, variant_ptr) * (intra ? piece_wgs : 1u)
                                     : (uint32_t)((((uint64_t)sub_points * plan.max_regular_bytes + 64u) + 255u) & ~uint64_t(255));
  const uint32_t segs_per_chunk = wide ? 1u : subs + 2u * n_adaptive;
  const uint64_t reg_stride = (uint64_t)subs * sub_stride;
  // WIDE: the slot takes the chunk's whole payload as one run -- the regular stream's worst case and, per adaptive field,
  // the largest section any mode can write (DeltaRle: 5 + 11 bytes per value)
  const uint64_t wide_slot = (((uint64_t)kPointsPerChunk * ((uint64_t)plan.max_regular_bytes + 11ull * n_adaptive) + 16ull * n_adaptive + 64ull) + 255ull) & ~255ull;
  const uint64_t slot_stride = wide ? wide_slot : reg_stride + (uint64_t)n_adaptive * kSectionStride;

  // one zero-filled block per call (one memset launch instead of three)
  const size_t z_anchor = 256;
  const size_t z_anchor2 = z_anchor + (((size_t)(n_chunks / 1024u + 1u) * 8u + 63u) & ~size_t(63));  // framing of the LZ4 blocks
  const size_t z_flags = z_anchor2 + (((size_t)(n_chunks / 1024u + 1u) * 8u + 63u) & ~size_t(63));
  const size_t z_segs = (z_flags + (size_t)n_chunks * std::max(1u, n_adaptive) + 63u) & ~size_t(63);
  const size_t z_bytes = z_segs + std::max<size_t>(16, (size_t)n_chunks * segs_per_chunk * sizeof(Seg));
  if ((rc = c->d_status.ensure(z_bytes)) != CLDN_HIP_OK) return rc;
  if ((rc = c->d_offsets.ensure((size_t)(n_clouds + 1) * sizeof(uint64_t))) != CLDN_HIP_OK) return rc;
  if ((rc = c->d_modes.ensure(std::max<size_t>(1, (size_t)n_clouds * std::max(1u, n_adaptive)))) != CLDN_HIP_OK)
    return rc;
  HIP_TRY(hipMemsetAsync(c->d_status.p, 0, z_bytes, c->stream));
  // a batch without a single point launches no probe: its clouds commit mode 0 (DeltaVarint), like an encode() call of
  // the reference that never reaches the analysis (src/v5_codec.cpp:934-949)
  if (n_chunks == 0 && n_clouds && n_adaptive) HIP_TRY(hipMemsetAsync(c->d_modes.p, 0, (size_t)n_clouds * n_adaptive, c->stream));

  const uint8_t* d_points = (const uint8_t*)points;
  uint8_t* d_outp = (uint8_t*)out;
  if (n_chunks) {
    if ((rc = c->d_payload.ensure((size_t)n_chunks * sizeof(uint32_t))) != CLDN_HIP_OK) return rc;
    if ((rc = c->d_dst.ensure((size_t)n_chunks * sizeof(uint64_t))) != CLDN_HIP_OK) return rc;
    if ((rc = c->d_slots.ensure(std::max<size_t>(16, (size_t)n_chunks * slot_stride))) != CLDN_HIP_OK) return rc;
    {  // look-back records of k_finish: tagged with the call's epoch, so only a fresh allocation is cleared
      const void* before = c->d_finrec.p;
      if ((rc = c->d_finrec.ensure((size_t)n_chunks * 16u + (size_t)n_chunks * 32u * 8u)) != CLDN_HIP_OK) return rc;  // rec, rec2, 32 workgroup records per chunk
      if (c->d_finrec.p != before) {
        HIP_TRY(hipMemsetAsync(c->d_finrec.p, 0, c->d_finrec.cap, c->stream));
        c->finish_epoch = 0;
      }
      if (++c->finish_epoch == 0u) {  // wrapped: old records could carry the new tags
        HIP_TRY(hipMemsetAsync(c->d_finrec.p, 0, c->d_finrec.cap, c->stream));
        c->finish_epoch = 1u;
      }
    }
    for (uint32_t a = 0; a < plan.n_adaptive; ++a) {  // (WIDE: plan.n_adaptive == 0, the columns live in the chunk scratch)
      if ((rc = c->d_cols[a].ensure((size_t)n_points * plan.adaptive[a].bpv + 64)) != CLDN_HIP_OK) return rc;
      if ((rc = c->d_ranks[a].ensure((size_t)n_points * 2 + 64)) != CLDN_HIP_OK) return rc;
    }
    if (wide) {
      if ((rc = c->d_wide_scratch.ensure((size_t)n_chunks * stage1_wide_scratch_bytes())) != CLDN_HIP_OK) return rc;
      const uint32_t ng = c->plan.n_gorilla_all;
      if (ng) {
        std::vector<void*> ptrs(ng);
        for (uint32_t g = 0; g < ng; ++g) {
          if ((rc = c->d_wide_pre[g].ensure((size_t)n_points * 16 + 64)) != CLDN_HIP_OK) return rc;
          ptrs[g] = c->d_wide_pre[g].p;
        }
        if ((rc = c->d_pre_ptrs.ensure(ptrs.size() * sizeof(void*))) != CLDN_HIP_OK) return rc;
        HIP_TRY(hipMemcpyAsync(c->d_pre_ptrs.p, ptrs.data(), ptrs.size() * sizeof(void*), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));  // `ptrs` lives on this stack frame
      }
    }
    if (plan.n_gorilla) {
      void* ptrs[kMaxGorilla] = {};
      for (uint32_t g = 0; g < plan.n_gorilla; ++g) {
        // 16-byte tokens per point (generic kernel) or one window word per piece (piece kernel, kGorWinStride per chunk)
        if ((rc = c->d_pre[g].ensure(std::max((size_t)n_points * 16, (size_t)n_chunks * 256) + 64)) != CLDN_HIP_OK) return rc;
        ptrs[g] = c->d_pre[g].p;
      }
      if ((rc = c->d_pre_ptrs.ensure(sizeof(ptrs))) != CLDN_HIP_OK) return rc;
      HIP_TRY(hipMemcpyAsync(c->d_pre_ptrs.p, ptrs, sizeof(ptrs), hipMemcpyHostToDevice, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));  // `ptrs` lives on this stack frame
    }
    if (points_loc == CLDN_HIP_HOST) {
      if ((rc = c->d_in.ensure((size_t)n_points * step)) != CLDN_HIP_OK) return rc;
      if (cloud_ptrs) {  // gather: every cloud straight from its own buffer to its place in the batch
        size_t at = 0;
        for (uint32_t k = 0; k < n_clouds; ++k) {
          const size_t bytes = (size_t)cloud_points[k] * step;
          if (bytes) HIP_TRY(hipMemcpyAsync((uint8_t*)c->d_in.p + at, cloud_ptrs[k], bytes, hipMemcpyHostToDevice, c->stream));
          at += bytes;
        }
      } else {
        HIP_TRY(hipMemcpyAsync(c->d_in.p, points, (size_t)n_points * step, hipMemcpyHostToDevice, c->stream));
      }
      d_points = (const uint8_t*)c->d_in.p;
    }
    if (out_loc == CLDN_HIP_HOST && !table) {
      if ((rc = c->d_out.ensure((size_t)need)) != CLDN_HIP_OK) return rc;
      d_outp = (uint8_t*)c->d_out.p;
    }
    if (lz4) {
      if ((rc = c->d_s1.ensure((size_t)need_s1)) != CLDN_HIP_OK) return rc;
      if ((rc = c->d_s1_offsets.ensure((size_t)(n_clouds + 1) * sizeof(uint64_t))) != CLDN_HIP_OK) return rc;
      if ((rc = c->d_payload2.ensure((size_t)n_chunks * sizeof(uint32_t))) != CLDN_HIP_OK) return rc;
      if ((rc = c->d_dst2.ensure((size_t)n_chunks * sizeof(uint64_t))) != CLDN_HIP_OK) return rc;
    }
  }
  EncodeLaunch L;
  memset(&L, 0, sizeof(L));
  L.plan = &plan;
  L.stream = c->stream;
  L.points = d_points;
  L.points_end = d_points + (size_t)n_points * step;
  L.chunks = (const ChunkDesc*)c->d_chunks.p;
  L.n_chunks = n_chunks;
  L.n_clouds = n_clouds;
  L.cloud_first_chunk = (const uint32_t*)c->d_cloud_first.p;
  L.slots = (uint8_t*)c->d_slots.p;
  L.slot_stride = slot_stride;
  L.reg_stride = reg_stride;
  L.subs = subs;
  L.sub_points = sub_points;
  L.sub_stride = sub_stride;
  L.segs = (Seg*)((uint8_t*)c->d_status.p + z_segs);
  L.segs_per_chunk = segs_per_chunk;
  for (uint32_t a = 0; a < plan.n_adaptive; ++a) {
    L.cols.p[a] = (uint8_t*)c->d_cols[a].p;
    L.ranks[a] = (uint16_t*)c->d_ranks[a].p;
  }
  if (wide) {
    L.wide = &c->wide_desc;
    L.wide_ops_host = c->plan.ops_all.data();
    L.wide_scratch = (uint8_t*)c->d_wide_scratch.p;
    L.wide_pre = (const uint4* const*)c->d_pre_ptrs.p;
  }
  for (uint32_t g = 0; g < plan.n_gorilla; ++g) L.pre.p[g] = (const uint4*)c->d_pre[g].p;
  L.pre_out = (uint4* const*)c->d_pre_ptrs.p;
  L.chunk_payload = (uint32_t*)c->d_payload.p;
  L.chunk_dst = (uint64_t*)c->d_dst.p;
  L.stream_offsets = lz4 ? (uint64_t*)c->d_s1_offsets.p : (uint64_t*)c->d_offsets.p;
  // device-resident outputs: the kernels write the caller's stream_offsets / chunk_sizes arrays themselves (no copy behind
  // the call: a device-to-device copy of a few bytes costs a launch, 3-5 us of a 50 us call)
  const bool direct_offsets = out_loc == CLDN_HIP_DEVICE && !lz4 && !table && stream_offsets != nullptr && ((uintptr_t)stream_offsets & 7u) == 0u;
  const bool direct_sizes = out_loc == CLDN_HIP_DEVICE && !lz4 && !table && chunk_sizes != nullptr && ((uintptr_t)chunk_sizes & 3u) == 0u;
  if (direct_offsets) L.stream_offsets = stream_offsets;
  if (direct_sizes) L.chunk_payload = chunk_sizes;
  L.modes = (uint8_t*)c->d_modes.p;
  L.modes_forced = false;
  if (!wide && c->last_modes_count && c->ev_last_modes && hipEventQuery(c->ev_last_modes) == hipSuccess) {
    // a copy of an earlier call's modes has landed: it becomes the hint until a newer one lands
    const uint8_t* lm = (const uint8_t*)c->h_last_modes.p;
    const uint32_t nf = c->last_modes_fields;
    for (uint32_t a = 0; a < (uint32_t)kMaxAdaptive; ++a) c->hint_cache[a] = 0;
    for (size_t k = 0; k < c->last_modes_count; ++k) {
      const uint8_t m = lm[k];
      c->hint_cache[k % nf] |= (m <= 3u) ? (uint8_t)(1u << m) : (uint8_t)0xF;
    }
    c->hint_valid = true;
    c->last_modes_count = 0;
  }
  for (uint32_t a = 0; a < (uint32_t)kMaxAdaptive; ++a) L.mode_hint[a] = c->hint_valid ? c->hint_cache[a] : (uint8_t)0xF;
  if (!c->forced_modes.empty() && n_adaptive && n_clouds) {
    for (uint32_t a = 0; a < plan.n_adaptive; ++a) L.mode_hint[a] = (uint8_t)(1u << c->forced_modes[a]);
    HIP_TRY(hipStreamSynchronize(c->stream));  // the previous call's upload from h_modes has to be over
    if ((rc = c->h_modes.ensure((size_t)n_clouds * n_adaptive)) != CLDN_HIP_OK) return rc;
    for (uint32_t k = 0; k < n_clouds; ++k)
      memcpy((uint8_t*)c->h_modes.p + (size_t)k * n_adaptive, c->forced_modes.data(), n_adaptive);
    HIP_TRY(hipMemcpyAsync(c->d_modes.p, c->h_modes.p, (size_t)n_clouds * n_adaptive, hipMemcpyHostToDevice,
                           c->stream));
    L.modes_forced = true;
  }
  L.fallback_flags = (uint8_t*)c->d_status.p + z_flags;
  L.fin_rec = (unsigned long long*)c->d_finrec.p;
  L.fin_rec2 = L.fin_rec + n_chunks;
  L.fin_anchor = (unsigned long long*)((uint8_t*)c->d_status.p + z_anchor);
  L.fin_epoch = c->finish_epoch;
  L.fin_ticket = (uint32_t*)c->d_status.p + 40;
  L.use_ticket = c->force_ticket ? 1u : 0u;
  L.test_timeout = c->test_timeout_once ? 1u : 0u;  // test hook (cldn_hip_debug_finish_timeout_once): this attempt reports a timeout
  c->test_timeout_once = false;
  L.chunks_only = table != nullptr;
  L.contiguous_flag = (uint32_t*)c->d_status.p + 42;
  if (pieces) {
    L.pieces = (const PieceDesc*)c->d_pieces.p;
    L.n_pieces = c->n_pieces;
    L.intra = intra;
    L.wgrec = L.fin_rec2 + n_chunks;
  }
  L.out = lz4 ? (uint8_t*)c->d_s1.p : d_outp;
  L.out_capacity = lz4 ? need_s1 : out_capacity;
  L.status = (uint32_t*)c->d_status.p;
  const size_t n_slots = c->slot_valid.size();
  const size_t slot = n_slots ? (size_t)(c->call_index % n_slots) : 0;
  L.events = n_slots ? &c->events[slot * 5] : nullptr;
  rc = stage1_launch_encode(L);
  if (rc != CLDN_HIP_OK) return rc;
  const void* d_sizes = c->d_payload.p;  // what chunk_sizes reports
  if (lz4 && n_chunks) {
    // stage 2 on the device: an LZ4 block per chunk payload (lz4_kernels.hip), written straight into the framed streams
    // every chunk has ceil(payload / 16 KiB) sub-ranges; the payloads of the batch are bounded by need_s1
    const uint32_t lz_sub = lz4_fast ? kLzFastSubBytes : kLzSubBytes, lz_mm = lz4_fast ? kLzFastMaxMatches : kLzMaxMatches;
    uint64_t max_subs = need_s1 / lz_sub + n_chunks;
    // the match lists take as many bytes as the payloads they describe: sized from the worst-case stage-1 bound that is
    // gigabytes for a large batch (32 x 1 M XYZI points: 1.3 GB for 207 MB of payload). Beyond 256 MB the call waits for
    // stage 1 and sizes them from the bytes it really wrote (one 8-byte read; the wait is ~5 % of what the LZ4 stage takes)
    bool stage1_failed = false;
    if (max_subs * lz_mm * sizeof(LzMatch) > (256ull << 20)) {
      uint64_t total_s1 = 0;
      uint32_t st1 = 0;  // the status word travels with the total: behind a stage 1 that gave up (ST_FINISH_TIMEOUT: k_finish
                         // returned before it wrote offsets, payload sizes and positions) the total is a stale number
      HIP_TRY(hipMemcpyAsync(&total_s1, (const uint64_t*)c->d_s1_offsets.p + n_clouds, sizeof(total_s1), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipMemcpyAsync(&st1, c->d_status.p, sizeof(st1), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      stage1_failed = st1 != 0u;
      if (!stage1_failed && total_s1 <= need_s1) max_subs = total_s1 / lz_sub + n_chunks;  // (every chunk's last sub-range may be partial)
    }
    if (stage1_failed) {  // no LZ4 stage over stale sizes: the status handling of the caller (ticket retry / error) takes it from here
      HIP_TRY(hipMemsetAsync(c->d_offsets.p, 0, (size_t)(n_clouds + 1) * sizeof(uint64_t), c->stream));
    } else {
      if ((rc = c->d_lz_matches.ensure((size_t)max_subs * lz_mm * sizeof(LzMatch))) != CLDN_HIP_OK) return rc;
      if ((rc = c->d_lz_counts.ensure((size_t)(max_subs * 7u + 2u * n_chunks + 1u) * sizeof(uint32_t))) != CLDN_HIP_OK) return rc;
      Lz4Launch Z;
      Z.stream = c->stream;
      Z.stage1 = (const uint8_t*)c->d_s1.p;
      Z.chunk_dst = (const uint64_t*)c->d_dst.p;
      Z.chunk_payload = (const uint32_t*)c->d_payload.p;
      Z.n_chunks = n_chunks;
      Z.fast = lz4_fast ? 1u : 0u;
      Z.max_subs = max_subs;
      Z.matches = (LzMatch*)c->d_lz_matches.p;
      Z.counts = (uint32_t*)c->d_lz_counts.p;
      Z.last_end = Z.counts + max_subs;
      Z.anchor_in = Z.last_end + max_subs;
      Z.sub_size = Z.anchor_in + max_subs;
      Z.sub_chunk = Z.sub_size + max_subs;
      Z.before = Z.sub_chunk + max_subs;
      Z.next_pos = Z.before + max_subs;
      Z.block_size = Z.next_pos + max_subs;
      Z.sub_first = Z.block_size + n_chunks;
      // the blocks go straight into the framed streams: [u32 size][block] per chunk, positions from a scan of the block sizes
      Z.cloud_first_chunk = (const uint32_t*)c->d_cloud_first.p;
      Z.n_clouds = n_clouds;
      Z.block_dst = (uint64_t*)c->d_dst2.p;
      Z.block_sizes_out = (uint32_t*)c->d_payload2.p;
      Z.stream_offsets = (uint64_t*)c->d_offsets.p;
      Z.out = d_outp;
      Z.out_capacity = out_capacity;
      Z.status = (uint32_t*)c->d_status.p;
      if ((rc = lz4_launch(Z)) != CLDN_HIP_OK) return rc;
      d_sizes = c->d_payload2.p;
    }
  } else if (lz4) {
    HIP_TRY(hipMemsetAsync(c->d_offsets.p, 0, (size_t)(n_clouds + 1) * sizeof(uint64_t), c->stream));
  }
  // remember this call's modes for the next call's launch hint (no synchronisation: the copy is only looked at
  // once its event has fired)
  // (a hint that is in place is refreshed with every 16th call only: it decides the launch shape, never a byte)
  if (!wide && n_adaptive && n_clouds && (size_t)n_clouds * n_adaptive <= 65536u && c->last_modes_count == 0 &&
      (!c->hint_valid || (c->call_index & 15u) == 0u)) {
    if (!c->ev_last_modes) HIP_TRY(hipEventCreateWithFlags(&c->ev_last_modes, hipEventDisableTiming));
    if (c->h_last_modes.ensure((size_t)n_clouds * n_adaptive) == CLDN_HIP_OK) {  // no copy in flight: buffer is free
      HIP_TRY(hipMemcpyAsync(c->h_last_modes.p, c->d_modes.p, (size_t)n_clouds * n_adaptive, hipMemcpyDeviceToHost,
                             c->stream));
      HIP_TRY(hipEventRecord(c->ev_last_modes, c->stream));
      c->last_modes_count = (size_t)n_clouds * n_adaptive;
      c->last_modes_fields = n_adaptive;
    }
  }
  if (n_slots) c->slot_valid[slot] = 1;
  ++c->call_index;
  if (table) {
    table->payload_base = (const uint8_t*)c->d_slots.p;
    table->chunk_stride = slot_stride;
    table->segments = (const cldn_hip_segment_t*)L.segs;
    table->segments_per_chunk = segs_per_chunk;
    table->chunk_sizes = (const uint32_t*)c->d_payload.p;
    table->not_contiguous = (const uint32_t*)c->d_status.p + 42;
    table->n_chunks = n_chunks;
    c->ct_valid = true;
    c->ct_n_chunks = n_chunks;
    c->ct_n_clouds = n_clouds;
    c->ct_slot_stride = slot_stride;
    c->ct_segs_per_chunk = segs_per_chunk;
    c->ct_segs_off = z_segs;
    c->ct_anchor_off = z_anchor;
    c->ct_need = need;
    if (modes && (size_t)n_clouds * n_adaptive)
      HIP_TRY(hipMemcpyAsync(modes, c->d_modes.p, (size_t)n_clouds * n_adaptive, hipMemcpyDeviceToDevice, c->stream));
    return CLDN_HIP_OK;
  }

  if (const char* dump = dev_env("CLDN_HIP_DEBUG_DUMP")) {  // diagnostics: segment table of the last call
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<Seg> hs((size_t)n_chunks * segs_per_chunk);
    std::vector<ChunkDesc> hc(n_chunks);
    if (n_chunks) {
      HIP_TRY(hipMemcpy(hs.data(), (uint8_t*)c->d_status.p + z_segs, hs.size() * sizeof(Seg), hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(hc.data(), c->d_chunks.p, hc.size() * sizeof(ChunkDesc), hipMemcpyDeviceToHost));
    }
    if (FILE* f = fopen(dump, "w")) {
      fprintf(f, "subs %u sub_points %u sub_stride %u segs_per_chunk %u slot_stride %llu\n", subs, sub_points, sub_stride,
              segs_per_chunk, (unsigned long long)slot_stride);
      for (uint32_t ci = 0; ci < n_chunks; ++ci) {
        fprintf(f, "chunk %u first %llu n %u cloud %u:", ci, (unsigned long long)hc[ci].first_point, hc[ci].n_points,
                hc[ci].cloud);
        for (uint32_t k = 0; k < segs_per_chunk; ++k)
          fprintf(f, " [%u,%u]", hs[(size_t)ci * segs_per_chunk + k].off, hs[(size_t)ci * segs_per_chunk + k].size);
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }

  const size_t modes_bytes = (size_t)n_clouds * n_adaptive;
  if (out_loc == CLDN_HIP_DEVICE) {
    if (stream_offsets && !direct_offsets)
      HIP_TRY(hipMemcpyAsync(stream_offsets, c->d_offsets.p, (size_t)(n_clouds + 1) * sizeof(uint64_t),
                             hipMemcpyDeviceToDevice, c->stream));
    if (chunk_sizes && n_chunks && !direct_sizes)
      HIP_TRY(hipMemcpyAsync(chunk_sizes, d_sizes, (size_t)n_chunks * sizeof(uint32_t),
                             hipMemcpyDeviceToDevice, c->stream));
    if (modes && modes_bytes)
      HIP_TRY(hipMemcpyAsync(modes, c->d_modes.p, modes_bytes, hipMemcpyDeviceToDevice, c->stream));
    return CLDN_HIP_OK;
  }

  // host outputs: read the sizes back first, then exactly the produced bytes
  if ((rc = c->h_result.ensure((size_t)(n_clouds + 1) * sizeof(uint64_t) + 64)) != CLDN_HIP_OK) return rc;
  uint64_t* h_off = (uint64_t*)c->h_result.p;
  uint32_t* h_status = (uint32_t*)((uint8_t*)c->h_result.p + (size_t)(n_clouds + 1) * sizeof(uint64_t));
  HIP_TRY(hipMemcpyAsync(h_off, c->d_offsets.p, (size_t)(n_clouds + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost,
                         c->stream));
  HIP_TRY(hipMemcpyAsync(h_status, c->d_status.p, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (*h_status & ST_FINISH_TIMEOUT) {
    // k_finish takes its workgroups in index order and waits for the records of those in front; a device that hands them
    // out in another order (or pre-empts them) makes the wait run into its bound. The ticket counter does not depend on
    // the order: the call is redone with it once, and the codec keeps it from then on.
    if (!c->force_ticket) {
      c->force_ticket = true;
      ++c->finish_retries;
      // the redone call takes this attempt's place in the timing slots (cldn_hip_codec_kernel_ms)
      --c->call_index;
      if (n_slots) c->slot_valid[slot] = 0;
      HIP_TRY(hipMemsetAsync(c->d_status.p, 0, sizeof(uint32_t), c->stream));
      return kRetryWithTicket;
    }
    return fail(CLDN_HIP_ERR_DEVICE, "k_finish: a workgroup waited too long for the sizes of the chunks before it (status 0x%x)", *h_status);
  }
  if (*h_status & ST_OUT_OVERFLOW)
    return fail(CLDN_HIP_ERR_CAPACITY, "Output buffer too small for uncompressed chunk");  // chunk_writer.cpp:34-36
  const uint64_t total = h_off[n_clouds];
  if (deferred) c->pending_total = total;
  else if (total) HIP_TRY(hipMemcpyAsync(out, d_outp, (size_t)total, hipMemcpyDeviceToHost, c->stream));
  if (chunk_sizes && n_chunks)
    HIP_TRY(hipMemcpyAsync(chunk_sizes, d_sizes, (size_t)n_chunks * sizeof(uint32_t), hipMemcpyDeviceToHost,
                           c->stream));
  if (modes && modes_bytes) HIP_TRY(hipMemcpyAsync(modes, c->d_modes.p, modes_bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (stream_offsets) memcpy(stream_offsets, h_off, (size_t)(n_clouds + 1) * sizeof(uint64_t));
  return CLDN_HIP_OK;
}

static int encode_stage1_impl(cldn_hip_codec_t* c, const void* points, int points_loc, const void* const* cloud_ptrs,
                              const uint64_t* cloud_points, uint32_t n_clouds, void* out, uint64_t out_capacity, int out_loc,
                              uint64_t* stream_offsets, uint32_t* chunk_sizes, uint8_t* modes,
                              cldn_hip_chunk_table_t* table = nullptr) {
  int rc = encode_stage1_once(c, points, points_loc, cloud_ptrs, cloud_points, n_clouds, out, out_capacity, out_loc, stream_offsets,
                              chunk_sizes, modes, table);
  if (rc == kRetryWithTicket)
    rc = encode_stage1_once(c, points, points_loc, cloud_ptrs, cloud_points, n_clouds, out, out_capacity, out_loc, stream_offsets,
                            chunk_sizes, modes, table);
  return rc;
}

extern "C" {

uint32_t cldn_hip_codec_finish_retries(const cldn_hip_codec_t* c) { return c ? c->finish_retries : 0u; }

int cldn_hip_encode_stage1(cldn_hip_codec_t* c, const void* points, int points_loc, const uint64_t* cloud_points,
                           uint32_t n_clouds, void* out, uint64_t out_capacity, int out_loc,
                           uint64_t* stream_offsets, uint32_t* chunk_sizes, uint8_t* modes) {
  return encode_stage1_impl(c, points, points_loc, nullptr, cloud_points, n_clouds, out, out_capacity, out_loc,
                            stream_offsets, chunk_sizes, modes);
}

int cldn_hip_encode_stage1_gather(cldn_hip_codec_t* c, const void* const* cloud_ptrs, const uint64_t* cloud_points,
                                  uint32_t n_clouds, void* out, uint64_t out_capacity, int out_loc,
                                  uint64_t* stream_offsets, uint32_t* chunk_sizes, uint8_t* modes) {
  if (n_clouds && !cloud_ptrs) return fail(CLDN_HIP_ERR_ARG, "cloud_ptrs is NULL");
  return encode_stage1_impl(c, nullptr, CLDN_HIP_HOST, cloud_ptrs, cloud_points, n_clouds, out, out_capacity, out_loc,
                            stream_offsets, chunk_sizes, modes);
}

int cldn_hip_encode_stage1_chunks(cldn_hip_codec_t* c, const void* points, int points_loc, const uint64_t* cloud_points,
                                  uint32_t n_clouds, cldn_hip_chunk_table_t* table, uint8_t* modes_device) {
  if (!table) return fail(CLDN_HIP_ERR_ARG, "table is NULL");
  return encode_stage1_impl(c, points, points_loc, nullptr, cloud_points, n_clouds, nullptr, 0, CLDN_HIP_DEVICE, nullptr, nullptr,
                            modes_device, table);
}

int cldn_hip_frame_chunks(cldn_hip_codec_t* c, void* out, uint64_t out_capacity, int out_loc, uint64_t* stream_offsets,
                          uint32_t* chunk_sizes) {
  if (!c) return fail(CLDN_HIP_ERR_ARG, "codec is NULL");
  if (!c->ct_valid) return fail(CLDN_HIP_ERR_ARG, "frame_chunks: no chunk table (cldn_hip_encode_stage1_chunks must be this codec's last encode call)");
  if (out_loc != CLDN_HIP_HOST && out_loc != CLDN_HIP_DEVICE) return fail(CLDN_HIP_ERR_ARG, "invalid memory location tag");
  if (out_capacity < c->ct_need)
    return fail(CLDN_HIP_ERR_CAPACITY, "Output buffer too small for worst-case compressed size (%llu < %llu)",
                (unsigned long long)out_capacity, (unsigned long long)c->ct_need);
  if (c->ct_need && !out) return fail(CLDN_HIP_ERR_ARG, "out is NULL");
  ENTER_DEVICE(c->device);
  int rc;
  const uint32_t n_chunks = c->ct_n_chunks, n_clouds = c->ct_n_clouds;
  uint8_t* d_outp = (uint8_t*)out;
  if (out_loc == CLDN_HIP_HOST) {
    c->pending_total = 0;
    if ((rc = c->d_out.ensure((size_t)std::max<uint64_t>(1, c->ct_need))) != CLDN_HIP_OK) return rc;
    d_outp = (uint8_t*)c->d_out.p;
  }
  if (++c->finish_epoch == 0u) {
    HIP_TRY(hipMemsetAsync(c->d_finrec.p, 0, c->d_finrec.cap, c->stream));
    c->finish_epoch = 1u;
  }
  // (anchors: zero since the encode call, or holding the values an earlier framing of the same table left -- the same ones)
  FrameLaunch F;
  F.stream = c->stream;
  F.chunks = (const ChunkDesc*)c->d_chunks.p;
  F.n_chunks = n_chunks;
  F.cloud_first_chunk = (const uint32_t*)c->d_cloud_first.p;
  F.n_clouds = n_clouds;
  F.slots = (const uint8_t*)c->d_slots.p;
  F.slot_stride = c->ct_slot_stride;
  F.segs = (const Seg*)((uint8_t*)c->d_status.p + c->ct_segs_off);
  F.segs_per_chunk = c->ct_segs_per_chunk;
  F.rec = (unsigned long long*)c->d_finrec.p;
  F.anchor = (unsigned long long*)((uint8_t*)c->d_status.p + c->ct_anchor_off);
  F.epoch = c->finish_epoch;
  F.ticket = (uint32_t*)c->d_status.p + 40;
  F.chunk_payload = (uint32_t*)c->d_payload.p;
  F.chunk_dst = (uint64_t*)c->d_dst.p;
  F.stream_offsets = (uint64_t*)c->d_offsets.p;
  F.out = d_outp;
  F.out_capacity = out_capacity;
  F.status = (uint32_t*)c->d_status.p;
  if ((rc = stage1_launch_frame(F)) != CLDN_HIP_OK) return rc;
  if (out_loc == CLDN_HIP_DEVICE) {
    if (stream_offsets)
      HIP_TRY(hipMemcpyAsync(stream_offsets, c->d_offsets.p, (size_t)(n_clouds + 1) * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
    if (chunk_sizes && n_chunks)
      HIP_TRY(hipMemcpyAsync(chunk_sizes, c->d_payload.p, (size_t)n_chunks * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    return CLDN_HIP_OK;
  }
  if ((rc = c->h_result.ensure((size_t)(n_clouds + 1) * sizeof(uint64_t) + 64)) != CLDN_HIP_OK) return rc;
  uint64_t* h_off = (uint64_t*)c->h_result.p;
  uint32_t* h_status = (uint32_t*)((uint8_t*)c->h_result.p + (size_t)(n_clouds + 1) * sizeof(uint64_t));
  HIP_TRY(hipMemcpyAsync(h_off, c->d_offsets.p, (size_t)(n_clouds + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(h_status, c->d_status.p, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (*h_status & ST_FINISH_TIMEOUT) return fail(CLDN_HIP_ERR_DEVICE, "k_finish: a workgroup waited too long (status 0x%x)", *h_status);
  if (*h_status & ST_OUT_OVERFLOW) return fail(CLDN_HIP_ERR_CAPACITY, "Output buffer too small for uncompressed chunk");
  const uint64_t total = h_off[n_clouds];
  if (total) HIP_TRY(hipMemcpyAsync(out, d_outp, (size_t)total, hipMemcpyDeviceToHost, c->stream));
  if (chunk_sizes && n_chunks)
    HIP_TRY(hipMemcpyAsync(chunk_sizes, c->d_payload.p, (size_t)n_chunks * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (stream_offsets) memcpy(stream_offsets, h_off, (size_t)(n_clouds + 1) * sizeof(uint64_t));
  return CLDN_HIP_OK;
}

int cldn_hip_viz_preprocess(cldn_hip_codec_t* c, const void* points, int points_loc, uint64_t n_points,
                            uint32_t point_step, uint32_t xyz_offset, float resolution, void* out, uint64_t out_capacity,
                            int out_loc, uint64_t* kept_points) {
  if (!c || !kept_points) return fail(CLDN_HIP_ERR_ARG, "viz_preprocess: NULL argument");
  *kept_points = 0;
  if ((points_loc != CLDN_HIP_HOST && points_loc != CLDN_HIP_DEVICE) ||
      (out_loc != CLDN_HIP_HOST && out_loc != CLDN_HIP_DEVICE))
    return fail(CLDN_HIP_ERR_ARG, "invalid memory location tag");
  if (point_step == 0 || (uint64_t)xyz_offset + 12u > point_step)
    return fail(CLDN_HIP_ERR_ARG, "viz_preprocess: the x/y/z triple does not fit the point (offset %u, step %u)", xyz_offset,
                point_step);
  if (!(resolution > 0.0f) || !std::isfinite(resolution))
    return fail(CLDN_HIP_ERR_ARG, "viz_preprocess: resolution must be positive and finite");
  if (n_points >= 0xffffffffull) return fail(CLDN_HIP_ERR_UNSUPPORTED, "viz_preprocess: more than 2^32 - 2 points");
  if (n_points == 0) return CLDN_HIP_OK;
  if (!points || !out) return fail(CLDN_HIP_ERR_ARG, "viz_preprocess: NULL buffer");
  const uint64_t bytes = n_points * point_step;
  if (out_capacity < bytes)
    return fail(CLDN_HIP_ERR_CAPACITY, "viz_preprocess: output needs room for every input point (%llu < %llu)",
                (unsigned long long)out_capacity, (unsigned long long)bytes);
  ENTER_DEVICE(c->device);
  int rc;
  const uint8_t* d_points = (const uint8_t*)points;
  if (points_loc == CLDN_HIP_HOST) {
    if ((rc = c->d_in.ensure((size_t)bytes)) != CLDN_HIP_OK) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_in.p, points, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    d_points = (const uint8_t*)c->d_in.p;
  }
  uint8_t* d_out = (uint8_t*)out;
  if (out_loc == CLDN_HIP_HOST) {
    c->pending_total = 0;  // the device output buffer is reused: a deferred encode output waiting in it is gone
    if ((rc = c->d_out.ensure((size_t)bytes)) != CLDN_HIP_OK) return rc;
    d_out = (uint8_t*)c->d_out.p;
  }
  const uint64_t cap = viz_table_capacity(n_points);
  if ((rc = c->d_viz_keys.ensure((size_t)cap * 16u)) != CLDN_HIP_OK) return rc;  // {key, first index, pad} per slot
  if ((rc = c->d_viz_slot.ensure((size_t)n_points * 4u)) != CLDN_HIP_OK) return rc;
  if ((rc = c->d_viz_blocks.ensure((size_t)((n_points + 1023u) / 1024u) * 4u + 16u)) != CLDN_HIP_OK) return rc;
  if ((rc = c->d_viz_total.ensure(16)) != CLDN_HIP_OK) return rc;
  VizLaunch L;
  L.stream = c->stream;
  L.points = d_points;
  L.n_points = n_points;
  L.point_step = point_step;
  L.xyz_offset = xyz_offset;
  L.inv_res = 1.0f / resolution;  // const float inv_res = 1.0f / xyz_res (ros_msg_utils.cpp:272)
  L.keys = (unsigned long long*)c->d_viz_keys.p;
  L.first = nullptr;  // (round 5: inside the table's entries)
  L.slot_of = (uint32_t*)c->d_viz_slot.p;
  L.block_count = (uint32_t*)c->d_viz_blocks.p;
  L.total = (unsigned long long*)c->d_viz_total.p;
  L.out = d_out;
  if ((rc = viz_launch(L)) != CLDN_HIP_OK) return rc;
  unsigned long long kept = 0;
  HIP_TRY(hipMemcpyAsync(&kept, c->d_viz_total.p, sizeof(kept), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (out_loc == CLDN_HIP_HOST && kept)
    HIP_TRY(hipMemcpy(out, d_out, (size_t)(kept * point_step), hipMemcpyDeviceToHost));
  *kept_points = kept;
  return CLDN_HIP_OK;
}

int cldn_hip_codec_set_decode_fill(cldn_hip_codec_t* c, int fill) {
  if (!c) return fail(CLDN_HIP_ERR_ARG, "codec is NULL");
  if (fill != CLDN_HIP_FILL_KEEP && fill != CLDN_HIP_FILL_ZERO) return fail(CLDN_HIP_ERR_ARG, "unknown decode fill %d", fill);
  c->decode_fill = fill;
  return CLDN_HIP_OK;
}

int cldn_hip_codec_decode_stats(cldn_hip_codec_t* c, uint32_t stats[4]) {
  if (!c || !stats) return fail(CLDN_HIP_ERR_ARG, "decode_stats: NULL argument");
  memset(stats, 0, 4 * sizeof(uint32_t));
  if (!c->d_status.p) return CLDN_HIP_OK;
  ENTER_DEVICE(c->device);
  HIP_TRY(hipMemcpyAsync(stats, (const uint32_t*)c->d_status.p + 8, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost,
                         c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return CLDN_HIP_OK;
}

int cldn_hip_codec_fetch_output(cldn_hip_codec_t* c, void* out, uint64_t out_capacity) {
  if (!c) return fail(CLDN_HIP_ERR_ARG, "codec is NULL");
  if (out_capacity < c->pending_total)
    return fail(CLDN_HIP_ERR_CAPACITY, "fetch_output: %llu bytes are waiting, the buffer holds %llu",
                (unsigned long long)c->pending_total, (unsigned long long)out_capacity);
  if (c->pending_total == 0) return CLDN_HIP_OK;
  if (!out) return fail(CLDN_HIP_ERR_ARG, "out is NULL");
  ENTER_DEVICE(c->device);
  HIP_TRY(hipMemcpyAsync(out, c->d_out.p, (size_t)c->pending_total, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->pending_total = 0;
  return CLDN_HIP_OK;
}

int cldn_hip_codec_pipeline(cldn_hip_codec_t* c, int mode, const void* points) {
  if (!c) return fail(CLDN_HIP_ERR_ARG, "codec is NULL");
  if (mode < 0 || mode > 2) return fail(CLDN_HIP_ERR_ARG, "pipeline mode %d out of range", mode);
  c->pipeline = mode;
  const uint8_t* p = (const uint8_t*)points;
  if (mode == 1 || stage1_piece_points(c->plan.dev, p) == 0u) return 1;
  return 2;
}

int cldn_hip_codec_force_modes(cldn_hip_codec_t* c, const uint8_t* modes, uint32_t n_modes) {
  if (!c) return fail(CLDN_HIP_ERR_ARG, "codec is NULL");
  if (!modes || n_modes == 0) {
    c->forced_modes.clear();
    return CLDN_HIP_OK;
  }
  if (n_modes != c->plan.n_adaptive_total())
    return fail(CLDN_HIP_ERR_ARG, "force_modes: %u modes given, the plan has %u adaptive fields", n_modes,
                c->plan.n_adaptive_total());
  for (uint32_t a = 0; a < n_modes; ++a)
    if (modes[a] > 3u) return fail(CLDN_HIP_ERR_ARG, "force_modes: invalid adaptive-int mode %u", modes[a]);
  c->forced_modes.assign(modes, modes + n_modes);
  return CLDN_HIP_OK;
}

int cldn_hip_decode_stage1(cldn_hip_codec_t* c, const void* streams, int streams_loc, const uint64_t* stream_offsets,
                           const uint64_t* cloud_points, uint32_t n_clouds, void* points_out, uint64_t out_capacity,
                           int out_loc) {
  return cldn_hip_decode_stage1_sized(c, streams, streams_loc, stream_offsets, cloud_points, n_clouds, nullptr, CLDN_HIP_HOST,
                                      points_out, out_capacity, out_loc);
}

int cldn_hip_decode_stage1_sized(cldn_hip_codec_t* c, const void* streams, int streams_loc, const uint64_t* stream_offsets,
                                 const uint64_t* cloud_points, uint32_t n_clouds, const uint32_t* chunk_sizes, int chunk_sizes_loc,
                                 void* points_out, uint64_t out_capacity, int out_loc) {
  if (!c) return fail(CLDN_HIP_ERR_ARG, "codec is NULL");
  if (chunk_sizes && chunk_sizes_loc != CLDN_HIP_HOST && chunk_sizes_loc != CLDN_HIP_DEVICE)
    return fail(CLDN_HIP_ERR_ARG, "invalid memory location tag");
  if (n_clouds && (!cloud_points || !stream_offsets)) return fail(CLDN_HIP_ERR_ARG, "NULL offsets / cloud_points");
  if ((streams_loc != CLDN_HIP_HOST && streams_loc != CLDN_HIP_DEVICE) ||
      (out_loc != CLDN_HIP_HOST && out_loc != CLDN_HIP_DEVICE))
    return fail(CLDN_HIP_ERR_ARG, "invalid memory location tag");
  ENTER_DEVICE(c->device);
  const DevPlan& plan = c->plan.dev;
  const uint32_t step = plan.point_step;
  if (n_clouds == 0) return CLDN_HIP_OK;

  uint64_t n_points = 0, n_chunks64 = 0;
  for (uint32_t k = 0; k < n_clouds; ++k) {
    if (stream_offsets[k + 1] < stream_offsets[k]) return fail(CLDN_HIP_ERR_ARG, "stream_offsets must be ascending");
    n_points += cloud_points[k];
    n_chunks64 += (cloud_points[k] + kPointsPerChunk - 1) / kPointsPerChunk;
  }
  if (n_chunks64 > 0x3fffffffull) return fail(CLDN_HIP_ERR_ARG, "batch too large");
  const uint32_t n_chunks = (uint32_t)n_chunks64;
  const uint64_t need = n_points * step;
  if (out_capacity < need) return fail(CLDN_HIP_ERR_CAPACITY, "Output buffer is too small to hold the decoded data");
  if (need && !points_out) return fail(CLDN_HIP_ERR_ARG, "points_out is NULL");
  const uint64_t stream_bytes = stream_offsets[n_clouds];
  if (stream_bytes && !streams) return fail(CLDN_HIP_ERR_ARG, "streams is NULL");

  int rc;
  if ((rc = c->d_status.ensure(256)) != CLDN_HIP_OK) return rc;
  HIP_TRY(hipMemsetAsync(c->d_status.p, 0, 256, c->stream));
  // tables: [stream_offsets u64 | cloud_first_point u64 | cloud_first_chunk u32] (n_clouds + 1 entries each)
  const size_t ne = (size_t)n_clouds + 1;
  const size_t table_bytes = ne * 8 + ne * 8 + ne * 4;
  // calls of a few clouds (one cloud per call: the subscriber's shape) pass the tables as a kernel argument: no upload
  const bool inline_tables = n_clouds <= kDecInlineClouds;
  uint64_t inl_so[kDecInlineClouds + 1], inl_fp[kDecInlineClouds + 1];
  uint32_t inl_fc[kDecInlineClouds + 1];
  uint32_t slot = 0;
  uint64_t* h_so = inl_so;
  uint64_t* h_fp = inl_fp;
  uint32_t* h_fc = inl_fc;
  PinnedBuf* stage_p = nullptr;
  if (!inline_tables) {
    slot = c->dec_stage_next++ % (uint32_t)cldn_hip_codec::kDecStageRing;
    PinnedBuf& stage = c->h_dec_stage[slot];
    if (c->dec_stage_ev[slot]) HIP_TRY(hipEventSynchronize(c->dec_stage_ev[slot]));  // its last upload has left the buffer
    else HIP_TRY(hipEventCreateWithFlags(&c->dec_stage_ev[slot], hipEventDisableTiming));
    if ((rc = stage.ensure(table_bytes)) != CLDN_HIP_OK) return rc;
    stage_p = &stage;
    h_so = (uint64_t*)stage.p;
    h_fp = h_so + ne;
    h_fc = (uint32_t*)(h_fp + ne);
  }
  const uint64_t base_off = stream_offsets[0];
  uint64_t fp = 0;
  uint32_t fc = 0;
  for (uint32_t k = 0; k < n_clouds; ++k) {
    h_so[k] = stream_offsets[k] - base_off;
    h_fp[k] = fp;
    h_fc[k] = fc;
    fp += cloud_points[k];
    fc += (uint32_t)((cloud_points[k] + kPointsPerChunk - 1) / kPointsPerChunk);
  }
  h_so[n_clouds] = stream_bytes - base_off;
  h_fp[n_clouds] = fp;
  h_fc[n_clouds] = fc;
  const size_t chunk_table_bytes = ((size_t)std::max(1u, n_chunks) * kDecChunkBytes + 63) & ~size_t(63);
  if ((rc = c->d_dec_meta.ensure(((table_bytes + 63) & ~size_t(63)) + chunk_table_bytes + (size_t)std::max(1u, n_chunks) * 18u + 64u)) != CLDN_HIP_OK)
    return rc;
  const bool dec_cols = c->plan.uses_v5 && plan.n_adaptive >= 1u && plan.n_adaptive <= 8u;
  for (uint32_t a = 0; a < 8u; ++a)
    if (dec_cols && a < plan.n_adaptive && plan.adaptive[a].bpv <= 4u &&
        (rc = c->d_dec_cols[a].ensure((size_t)n_points * plan.adaptive[a].bpv + 64)) != CLDN_HIP_OK)
      return rc;
  uint8_t* meta = (uint8_t*)c->d_dec_meta.p;
  if (!inline_tables) {
    HIP_TRY(hipMemcpyAsync(meta, stage_p->p, table_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipEventRecord(c->dec_stage_ev[slot], c->stream));
  }

  const uint8_t* d_streams = (const uint8_t*)streams + base_off;
  if (streams_loc == CLDN_HIP_HOST) {
    if ((rc = c->d_in.ensure((size_t)std::max<uint64_t>(1, stream_bytes - base_off))) != CLDN_HIP_OK) return rc;
    if (stream_bytes > base_off)
      HIP_TRY(hipMemcpyAsync(c->d_in.p, (const uint8_t*)streams + base_off, (size_t)(stream_bytes - base_off),
                             hipMemcpyHostToDevice, c->stream));
    d_streams = (const uint8_t*)c->d_in.p;
  }
  uint8_t* d_outp = (uint8_t*)points_out;
  if (out_loc == CLDN_HIP_HOST) {
    c->pending_total = 0;  // the device output buffer is reused: a deferred encode output waiting in it is gone
    if ((rc = c->d_out.ensure((size_t)std::max<uint64_t>(1, need))) != CLDN_HIP_OK) return rc;
    d_outp = (uint8_t*)c->d_out.p;
    // bytes of a point that no field covers keep the caller's content (src/field_decoder.cpp:72-76): bring it along
    if (need && c->plan.has_padding) {
      if (c->decode_fill == CLDN_HIP_FILL_ZERO) HIP_TRY(hipMemsetAsync(d_outp, 0, (size_t)need, c->stream));
      else HIP_TRY(hipMemcpyAsync(d_outp, points_out, (size_t)need, hipMemcpyHostToDevice, c->stream));
    }
  }

  DecodeLaunch L;
  memset(&L, 0, sizeof(L));
  L.plan = &plan;
  L.stream = c->stream;
  L.uses_v5 = c->plan.uses_v5 ? 1u : 0u;
  L.streams = d_streams;
  L.stream_offsets = (const uint64_t*)meta;
  L.cloud_first_point = (const uint64_t*)(meta + ne * 8);
  L.cloud_first_chunk = (const uint32_t*)(meta + ne * 16);
  if (inline_tables) {
    L.h_stream_offsets = h_so;
    L.h_cloud_first_point = h_fp;
    L.h_cloud_first_chunk = h_fc;
  }
  L.n_clouds = n_clouds;
  L.n_chunks = n_chunks;
  L.chunks = meta + ((table_bytes + 63) & ~size_t(63));
  L.reg_end = (uint32_t*)(meta + ((table_bytes + 63) & ~size_t(63)) + chunk_table_bytes);
  L.reg_end_pre = L.reg_end + std::max(1u, n_chunks);
  L.sec_done = (uint8_t*)(L.reg_end_pre + std::max(1u, n_chunks));
  L.sec_cols = L.sec_done + std::max(1u, n_chunks);
  L.chunk_sizes = nullptr;
  uint32_t* const d_sizes = (uint32_t*)(((uintptr_t)(L.sec_cols + std::max(1u, n_chunks)) + 3u) & ~uintptr_t(3));
  L.slices_done = d_sizes + std::max(1u, n_chunks);
  if (dec_cols && plan.n_adaptive == 1u && n_chunks) {  // slice records of k_sections_cols_fast
    const void* before = c->d_dec_rec.p;
    if ((rc = c->d_dec_rec.ensure((size_t)n_chunks * 48u * 16u)) != CLDN_HIP_OK) return rc;
    if (c->d_dec_rec.p != before || ++c->dec_epoch == 0u) {  // fresh memory, or the tags have wrapped
      HIP_TRY(hipMemsetAsync(c->d_dec_rec.p, 0, c->d_dec_rec.cap, c->stream));
      c->dec_epoch = 1u;
    }
    L.slice_rec = (unsigned long long*)c->d_dec_rec.p;
    L.slice_epoch = c->dec_epoch;
  }
  if (chunk_sizes && n_chunks) {
    if (chunk_sizes_loc == CLDN_HIP_DEVICE) {
      L.chunk_sizes = chunk_sizes;
    } else {  // (pageable source: the copy is over when the call returns)
      HIP_TRY(hipMemcpyAsync(d_sizes, chunk_sizes, (size_t)n_chunks * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
      L.chunk_sizes = d_sizes;
    }
  }
  for (uint32_t a = 0; a < 8u; ++a)
    L.cols[a] = (dec_cols && a < plan.n_adaptive && plan.adaptive[a].bpv <= 4u) ? (uint8_t*)c->d_dec_cols[a].p : nullptr;
  L.dsec = nullptr;
  L.secs_ok = nullptr;
  L.done_cnt = nullptr;
  if (c->plan.uses_v5 && plan.n_adaptive >= 1u && plan.n_adaptive <= 8u && n_chunks) {  // sections side by side (stage1_decode_sections_w.h)
    const size_t rows = (size_t)plan.n_adaptive * n_chunks * kDecChunkBytes;
    if ((rc = c->d_dec_secs.ensure(rows + (size_t)n_chunks * 8u + 256u)) != CLDN_HIP_OK) return rc;
    L.dsec = c->d_dec_secs.p;
    L.done_cnt = (uint32_t*)((uint8_t*)c->d_dec_secs.p + ((rows + 63u) & ~size_t(63)));
    L.secs_ok = (uint8_t*)(L.done_cnt + n_chunks);
  }
  if (plan.varint_and_raw && n_chunks) {  // one bit per stream byte + a word per chunk (k_mark_token_ends)
    if ((rc = c->d_dec_bits.ensure((size_t)((stream_bytes - base_off) / 8u) + (size_t)n_chunks * 4u + 256u)) != CLDN_HIP_OK) return rc;
    L.token_ends = (uint32_t*)c->d_dec_bits.p;
  }
  L.out = d_outp;
  L.fill_zero = c->decode_fill == CLDN_HIP_FILL_ZERO ? 1u : 0u;
  L.status = (uint32_t*)c->d_status.p;
  if (c->plan.wide) {  // the serial decoder with the plan in device memory; its per-op state: 16 bytes per op and chunk
    if ((rc = c->d_wide_state.ensure((size_t)std::max(1u, n_chunks) * std::max<size_t>(1, c->plan.ops_all.size()) * 16u)) != CLDN_HIP_OK) return rc;
    L.wide = &c->wide_desc;
    L.wide_state = c->d_wide_state.p;
  }
  if (c->dec_stats_chunks && c->ev_dec_stats && hipEventQuery(c->ev_dec_stats) == hipSuccess) {  // an earlier call's counters have landed
    const uint32_t* hs = (const uint32_t*)c->h_dec_stats.p;
    c->dec_palette_hint = hs[4] == c->dec_stats_chunks && hs[2] == 0u && hs[3] == 0u;  // all folded by the guess, nothing serial
    // (kStatDvChunks = word 13, kStatDvMode = word 14: no chunk with a DeltaVarint section / every chunk's decoded by k_section_dv_w)
    c->dec_dv_hint = hs[6] == 0u ? 1u : (hs[5] == c->dec_stats_chunks && hs[2] == 0u && hs[3] == 0u ? 2u : 0u);
    c->dec_stats_chunks = 0;
    c->dec_stats_seen = true;
  }
  L.palette_hint = c->dec_palette_hint ? 1u : 0u;
  L.dv_hint = c->dec_dv_hint;
  // small batches: the point kernel's pieces spread over several workgroups per chunk (SPLIT launches) want their workspace
  L.wp_parts = c->test_split_parts == 0u ? wp_split_parts(n_chunks) : c->test_split_parts;
  if (!c->plan.wide && L.wp_parts > 1u) {
    uint64_t chunk_bound = (uint64_t)kPointsPerChunk * c->plan.ref_max_point_bytes;
    if (c->plan.uses_v5) chunk_bound += (uint64_t)c->plan.fields.size() * 32u + 1024u;
    const uint64_t maxp = chunk_bound / 992u + 3u;  // (kWpPiece, stage1_decode_wave.h: pieces of 62 units)
    if (maxp <= 4096u) {
      if ((rc = c->d_dec_split.ensure(wp_split_bytes(n_chunks, (uint32_t)maxp))) != CLDN_HIP_OK) return rc;
      L.wp_split = c->d_dec_split.p;
      L.wp_maxp = (uint32_t)maxp;
    }
  }
  L.events = c->dec_events[0] ? c->dec_events : nullptr;
  if (L.events) {
    (void)hipEventRecord(L.events[0], c->stream);
    (void)hipEventRecord(L.events[1], c->stream);  // (recorded again in front of the regular-stream kernel, if the call has one)
    (void)hipEventRecord(L.events[2], c->stream);
  }
  ++c->dec_call_index;
  if ((rc = stage1_launch_decode(L)) != CLDN_HIP_OK) return rc;
  if (L.events) {
    (void)hipEventRecord(L.events[3], c->stream);
    c->dec_events_valid = true;
  }
  // (the counters of every 4th call are enough once one copy has landed: the hints pick a launch shape, never a byte)
  if (n_chunks && c->plan.uses_v5 && c->dec_stats_chunks == 0 && (!c->dec_stats_seen || (c->dec_call_index & 3u) == 0u) &&
      c->h_dec_stats.ensure(64) == CLDN_HIP_OK) {  // (no copy in flight)
    if (!c->ev_dec_stats) HIP_TRY(hipEventCreateWithFlags(&c->ev_dec_stats, hipEventDisableTiming));
    HIP_TRY(hipMemcpyAsync(c->h_dec_stats.p, (const uint32_t*)c->d_status.p + 8, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(c->ev_dec_stats, c->stream));
    c->dec_stats_chunks = n_chunks;
  }

  if (out_loc == CLDN_HIP_DEVICE) return CLDN_HIP_OK;
  uint32_t st = 0;
  HIP_TRY(hipMemcpyAsync(&st, c->d_status.p, sizeof(st), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (st & ST_CORRUPT) return fail(CLDN_HIP_ERR_CORRUPT, "malformed stage-1 stream (truncated, bad chunk size, bad mode or trailing bytes)");
  if (need) HIP_TRY(hipMemcpy(points_out, d_outp, (size_t)need, hipMemcpyDeviceToHost));
  return CLDN_HIP_OK;
}

int cldn_hip_decode_stage1_unframed(cldn_hip_codec_t* c, const void* payload, uint64_t size, int payload_loc,
                                    void* points_out, uint64_t out_capacity, int out_loc) {
  if (!c) return fail(CLDN_HIP_ERR_ARG, "codec is NULL");
  if ((payload_loc != CLDN_HIP_HOST && payload_loc != CLDN_HIP_DEVICE) || (out_loc != CLDN_HIP_HOST && out_loc != CLDN_HIP_DEVICE))
    return fail(CLDN_HIP_ERR_ARG, "invalid memory location tag");
  if (c->plan.uses_v5) return fail(CLDN_HIP_ERR_ARG, "unframed streams (wire version 2) never use the V5 codec");
  if (size == 0) return CLDN_HIP_OK;
  if (!payload) return fail(CLDN_HIP_ERR_ARG, "payload is NULL");
  if (size > 0xffffffffull) return fail(CLDN_HIP_ERR_UNSUPPORTED, "unframed payload of more than 4 GiB");
  ENTER_DEVICE(c->device);
  const uint32_t step = c->plan.dev.point_step;
  const uint64_t cap_points = std::min<uint64_t>(out_capacity / step, 0xffffffffull);
  int rc;
  if ((rc = c->d_status.ensure(256)) != CLDN_HIP_OK) return rc;
  HIP_TRY(hipMemsetAsync(c->d_status.p, 0, 256, c->stream));
  if ((rc = c->d_dec_meta.ensure(256)) != CLDN_HIP_OK) return rc;
  const uint8_t* d_payload = (const uint8_t*)payload;
  if (payload_loc == CLDN_HIP_HOST) {
    if ((rc = c->d_in.ensure((size_t)size)) != CLDN_HIP_OK) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_in.p, payload, (size_t)size, hipMemcpyHostToDevice, c->stream));
    d_payload = (const uint8_t*)c->d_in.p;
  }
  uint8_t* d_outp = (uint8_t*)points_out;
  const uint64_t out_bytes = cap_points * step;
  if (out_loc == CLDN_HIP_HOST) {
    c->pending_total = 0;
    if ((rc = c->d_out.ensure((size_t)std::max<uint64_t>(1, out_bytes))) != CLDN_HIP_OK) return rc;
    d_outp = (uint8_t*)c->d_out.p;
    // how many points the stream holds is not known beforehand: every byte the decoder does not write keeps the caller's
    // content (uncovered bytes of a point, and all points behind the last decoded one)
    if (out_bytes) HIP_TRY(hipMemcpyAsync(d_outp, points_out, (size_t)out_bytes, hipMemcpyHostToDevice, c->stream));
  }
  if (c->plan.wide && (rc = c->d_wide_state.ensure(std::max<size_t>(1, c->plan.ops_all.size()) * 16u)) != CLDN_HIP_OK) return rc;
  if ((rc = stage1_launch_decode_unframed(c->plan.dev, c->stream, d_payload, (uint32_t)size, (uint32_t)cap_points, c->d_dec_meta.p,
                                          d_outp, (uint32_t*)c->d_status.p, c->plan.wide ? &c->wide_desc : nullptr,
                                          c->d_wide_state.p)) != CLDN_HIP_OK)
    return rc;
  if (out_loc == CLDN_HIP_DEVICE) return CLDN_HIP_OK;
  uint32_t st = 0;
  HIP_TRY(hipMemcpyAsync(&st, c->d_status.p, sizeof(st), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (st & ST_CORRUPT) return fail(CLDN_HIP_ERR_CORRUPT, "malformed stage-1 stream (truncated point, or more points than the output holds)");
  if (out_bytes) HIP_TRY(hipMemcpy(points_out, d_outp, (size_t)out_bytes, hipMemcpyDeviceToHost));
  return CLDN_HIP_OK;
}

}  // extern "C"
