// plan.hpp -- internal interface of the bucket plan (bucket_plan.hip) and of the updates that consume it
// (plan_update.hip); the C ABI on top of it is in include/rechorus_hip.h (rc_bucket_plan, rc_plan_update).
#pragma once
#include "common.hpp"
#include "opt_math.hpp"

namespace rc {

constexpr int kPlanTile = 8192;        // positions per workgroup of the count / scatter kernels
constexpr int kPlanThreads = 512;
constexpr int kPlanWaves = kPlanThreads / 64;
constexpr int kPlanRounds = kPlanTile / kPlanThreads;  // keys per thread
constexpr int kPlanMaxBuckets = 4096;
constexpr int kPlanMaxShift = 13;      // ids per bucket <= 8192 (one 32 KiB LDS cursor table per wave)
constexpr int kPlanMinShift = 8;
constexpr int kPlanLongSeg = 32;       // occurrences one lane-group sums sequentially
constexpr int kPlanChunk = 256;        // occurrences per workgroup for hot rows

struct PlanGeom {
  int ok;
  int shift;             // ids per bucket = 1 << shift
  uint32_t nb_a, nb_b, nb;
  uint32_t base_b;       // key of list b = base_b + id   (a multiple of the bucket width)
  uint32_t tiles;
  int bucket_bits;       // bits of a bucket index
  int64_t n;
  // hashed buckets (sparse or very wide id spaces, where one bucket per 8,192-id range would cost more table cells than
  // there are keys): bucket = top bits of id * 0x9E3779B1 (nb_a, nb_b powers of two, list b behind list a), the id itself
  // is carried as the in-bucket key and grouped through an LDS hash table (plan_bucket_hash_kernel)
  int hashed;
  int log_nb_a, log_nb_b;
  // narrow id-range buckets (shift <= 5: at most 32 ids per bucket) for DENSE batches -- several keys per id of the table,
  // e.g. 0.6 M candidate + history occurrences over SASRec's 8.7 K items: the partition alone almost groups by row, and
  // one WAVE per bucket finishes the job with ballot ranks (plan_bucket_narrow_kernel): no LDS atomics, so a row with
  // tens of thousands of occurrences costs what its keys cost to stream, not a serialised atomic each
  int narrow;
};
constexpr int kPlanNarrowMaxShift = 5;
constexpr uint32_t kPlanHashSlots = 8192;   // LDS hash table of a hashed bucket (keys + cells: 64 KB)
constexpr int kPlanHashKeysPerBucket = 1024; // sizing target: distinct rows per bucket <= keys per bucket ~ load 1/8
// want: -1 = by density (hashed buckets where the id range is more than 16 cells per key, narrow id-range buckets where a
// batch brings >= 4 keys per id, wide id-range buckets between), 0 = id-range buckets (wide or narrow) wherever they exist,
// 1 = hashed, 2 = wide id-range buckets or nothing (callers that need the id-indexed bitmap)
PlanGeom plan_geometry(int64_t n_a, int64_t n_b, int64_t range_a, int64_t range_b, int want = -1);

// counters (device uint32[PC_N]); zeroed by the first kernel of every plan
enum { PC_ROWS_A = 0, PC_ROWS_B = 1, PC_LONG = 2, PC_CHUNKS = 3, PC_STATUS = 4, PC_N = 8 };

struct PlanWs {            // carved from the caller's workspace
  uint32_t* hist;          // [nb][tiles] per-tile bucket counts -> exclusive prefixes
  uint32_t* totals;        // [nb]
  uint32_t* bucket_base;   // [nb + 1]
  uint16_t* lid;           // [n] bucketed keys, split by consumer: id within its bucket (< 2^13) ...
  uint32_t* pos;           // [n] ... and batch position (the singleton bitmap and the counting passes read lid only)
  uint32_t* lid32;         // [n] hashed geometry: the whole id (aliases nothing; lid is unused then)
  uint32_t* counters;      // PC_N
  size_t total;
};
PlanWs carve_plan_ws(void* base, int64_t n);

struct PlanLongRow { uint32_t row, start, n, cbase, nchunks, side, pad0, pad1; };
struct PlanChunkInfo { uint32_t lrow, k; };

struct PlanLongWs {
  PlanLongRow* lrows;
  PlanChunkInfo* chunks;
  float* partial;
  uint32_t long_cap, chunk_cap;
  size_t total;
};
PlanLongWs carve_plan_long_ws(void* base, int64_t n, int d);

struct PlanArgs {
  const int64_t* ids_a;
  const int64_t* ids_b;
  uint32_t n_a, n;
  int64_t range_a, range_b;
  PlanGeom g;
  PlanWs w;
  int list_single_a;       // 0: single-occurrence rows of list a are flagged, not listed
  uint8_t* single_a;       // [round_up(n_a, kPlanTile)] or null
  rc_plan_row* rows_a;
  rc_plan_row* rows_b;
  uint32_t* n_rows_a;      // device counters (may point into w.counters)
  uint32_t* n_rows_b;
  uint32_t* occ;           // [n]
  int flags_done;          // the singleton information exists already (bitmap_a from plan_launch_front): the bucket kernel skips single_a
  uint32_t* bitmap_a;      // [nb_a << (shift - 5)] or null: bit (id) = 1 iff row id of list a occurs at least twice in the batch
  // hot rows (more than kPlanLongSeg occurrences) of the listed rows: registered by the bucket kernel itself when
  // emit_long is set (long-row records + chunk list in lw, counters PC_LONG / PC_CHUNKS), so that the consumer can
  // reduce their chunks in the SAME launch as the short rows (plan_update.hip, BPRMF step); otherwise the row-update
  // kernel finds them while it walks the records
  int emit_long;
  // the bucket kernel writes bitmap_a itself (it holds every id's count after its first pass): a plan that is prepared as a
  // whole -- the look-ahead plan of the next batch -- needs no separate bitmap launch (a second zero + count pass per bucket)
  int bitmap_in_bucket;
  PlanLongWs lw;
};
int plan_launch(const PlanArgs& a, hipStream_t s, hipEvent_t* ev_after_scatter);
// the same plan in two parts, so that a caller can run `back` on another stream (train_step.hip);
// bitmap: also fill a.bitmap_a (what the fused BPRMF kernel needs from the plan)
int plan_launch_front(const PlanArgs& a, bool bitmap, hipStream_t s);
constexpr size_t kPlanBitmapWords = (size_t)kPlanMaxBuckets << (kPlanMaxShift - 5);   // capacity that fits every geometry
int plan_launch_back(const PlanArgs& a, hipStream_t s);

// ---- consumers --------------------------------------------------------------------------------------
struct PlanTable { float* W; float* M; float* V; };

// gradient row of occurrence o:  o < n_split: coef[o] * src[src_index ? src_index[o / div] : o / div]
//                                else          src2[o - n_split]
struct PlanGrad {
  const float* coef;
  const float* src;
  const int64_t* src_index;
  int div;
  const float* src2;
  uint32_t n_split;
  // pair mode (two tables of width D/2 that share their ids, e.g. NeuMF's mf / mlp embedding of a side): the row of
  // width D is [table a | table b]; lanes of the lower half read src2 / update table a, the upper half src2b / table b
  const float* src2b;
  int pair;
  uint32_t pair_ld4;   // pair mode: row stride of src2 / src2b in float4 units (0: D / 8, two contiguous [n, D / 2] arrays) -- both
                       // halves of one [n, ld] block, e.g. the (d mf | d mlp) gradient rows a sharded NeuMF rank receives
};


int device_cus();

}  // namespace rc
