// oracle/restate/orc_common.h -- TEST INFRASTRUCTURE: shared helpers of the CPU restatement.
// The restatement is a plain, scalar re-expression of the reference's arithmetic for the hot path;
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it.  It is never
// linked into or called from the product library (vvcsoftware_vtm_amd/csrc).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include "vvcgpu.h"   // shares the parameter structs of the C ABI so tests use one layout

typedef int16_t Pel;
typedef int32_t TCoeff;

static inline int clip3i(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int sgni(int v) { return (0 < v) - (v < 0); }
#define ORC_API extern "C" __attribute__((visibility("default")))
