// krisp_hip.hip -- hand-written HIP (gfx950 / MI355X) for krisp_fasta's hot path:
// 2-bit pack -> both-strand keys -> MSD radix partition (LDS digit histograms,
// per-workgroup private cursors) -> LDS bucket sort -> n-way intersection with
// diagnostic-column masks -> candidate compaction -> record collection.
// C ABI: include/krisp_hip.h (each entry point cites the reference seam it replaces).
//
// Integer / byte work, HBM-bound: no MFMA anywhere (DESIGN.md "kernels").
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>       // the one exchange step of the multi-GPU flow (h_comm.inc)
#include <dlfcn.h>
#include <sys/stat.h>
#include <zlib.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <condition_variable>
#include <cerrno>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "krisp_hip.h"

// One translation unit, cut into parts by subject (kernels k_*, host code h_*):
#include "k_keys.inc"        // typedefs, key geometry, tunables, key generators
#include "k_sort.inc"        // pack, histograms, pass 0 / 1 / 2, chunk table, LDS sort, merge fallback
#include "k_sort2.inc"       // pass 1 / pass 2 / LDS sort as persistent, software-pipelined kernels (buffer loads, counted waits)
#include "k_intersect.inc"   // n-way intersection, candidate compaction, collection, list merge
#include "k_intersect3.inc"  // the same intersection as a persistent, software-pipelined kernel (items of whole buckets)
#include "k_intersect3t.inc" // ... with 32-bit heads where the geometry allows
#include "k_text.inc"        // the reference reader on the device: file text -> upload buffer
#include "k_inflate.inc"     // BGZF members inflated on the device, a lane per member (round 6)
#include "k_wide.inc"        // wide path kernels, copy kernel

#include "h_core.inc"        // context, buffers, parameters, upload, sort, finalize   (opens extern "C")
#include "h_intersect.inc"   // kr_intersect, candidate lists, kr_collect
#include "h_wide.inc"        // kr_wide_run
#include "h_pgzip.inc"       // one gzip member inflated on several threads (host only)
#include "h_ingest.inc"      // file -> inflate -> parse -> pinned upload buffer (host side)
#include "h_comm.inc"        // multi-GPU exchange: RCCL (or files, for rehearsal) tree reduction of candidates, gather of records
#include "h_text.inc"        // FASTA text parser, IUPAC side-channel scan (host only, no HIP)
#include "h_misc.inc"        // timers, debug entries, FASTA text parser

}  // extern "C"
