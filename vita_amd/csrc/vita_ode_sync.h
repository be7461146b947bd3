// Hand-off state shared by the in-launch samplers (vita_ode_persistent.hip: teams at chip-filling batches;
// vita_ode_resident.hip: the whole grid at serving batches).  One region of the sampler's workspace:
//   word  kTeamStride * i          arrival counter i, alone on its 128-byte line (teams, or the grid barrier's eight slots)
//   word  kTeamStride * kMaxTeams  timeout word of the running launch (zeroed before every launch)
//   word  kStickyWord              first timeout code of the whole call = what vita_ode_sample_status reports
#pragma once
#include "vita_common.h"

namespace vita {

constexpr int kTeamStride = 32;          // counters 128 bytes apart
constexpr int kMaxTeams = 64;
constexpr int kSyncWords = kTeamStride * (kMaxTeams + 1);   // + the timeout word's line (zeroed before every launch)
constexpr int kStickyWord = kSyncWords;                      // first timeout code of the whole call (zeroed once per call)
inline size_t ode_sync_words_bytes() { return (size_t)(kSyncWords + kTeamStride) * sizeof(unsigned); }

__device__ __forceinline__ unsigned long long clock64_shader() {   // shader clock (follows DVFS)
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ unsigned long long clock64_real() {     // 100 MHz constant clock
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

}  // namespace vita
