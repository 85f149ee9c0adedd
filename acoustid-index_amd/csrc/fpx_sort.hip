// fpx_sort.hip -- device radix sort of 64-bit keys used by the batch pipeline
// (query (hash,q) pairs, (q,doc) hit records, candidate keys).  Thin wrapper over
// rocPRIM's device radix sort so the slow-to-compile header stays in its own TU.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "fpx_internal.h"

namespace fpx {

// rocPRIM switches to a merge sort (dozens of small launches) below 2^20 keys by default; the batch pipeline sorts
// 10^5..10^6 keys per stage at small batch sizes, where the Onesweep radix sort (one histogram + one launch per 8 bits)
// is several times faster.
// rocPRIM 4.2 carries no tuned Onesweep configuration for gfx950 (it falls back to 512 threads x 12 keys); measured on the
// batch's two big sorts (8.2 M pair keys, 51 M hit records): 1024 x 8 is 7 % faster than that, 512 x 8 / 256 x 16 /
// 1024 x 6 are 15 - 35 % slower.  The histogram kernel likes more keys per thread: 1024 x 16 saves another 4 %.
using OnesweepConfig = rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 16>, rocprim::kernel_config<1024, 8>, 8,
                                                           rocprim::block_radix_rank_algorithm::match>;
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, OnesweepConfig, 16384>;

size_t sort_u64_temp_bytes(size_t n, unsigned begin_bit, unsigned end_bit)
{
    size_t bytes = 0;
    rocprim::double_buffer<uint64_t> db(nullptr, nullptr);
    (void)rocprim::radix_sort_keys<SortConfig>(nullptr, bytes, db, n, begin_bit, end_bit, (hipStream_t)0);
    return bytes;
}

// Sorts n keys held in bufs[0]; returns the index (0/1) of the buffer that holds the result.
hipError_t sort_u64(void* temp, size_t temp_bytes, uint64_t* buf0, uint64_t* buf1, size_t n,
                    unsigned begin_bit, unsigned end_bit, hipStream_t stream, int* result_in)
{
    if (n == 0) { *result_in = 0; return hipSuccess; }
    if (end_bit <= begin_bit) { *result_in = 0; return hipSuccess; }
    rocprim::double_buffer<uint64_t> db(buf0, buf1);
    hipError_t e = rocprim::radix_sort_keys<SortConfig>(temp, temp_bytes, db, n, begin_bit, end_bit, stream);
    *result_in = (db.current() == buf0) ? 0 : 1;
    return e;
}

// stream compaction of 64-bit items by a byte flag (segment merges drop the items of superseded docs)
size_t select_u64_temp_bytes(size_t n)
{
    size_t bytes = 0;
    (void)rocprim::select(nullptr, bytes, (const uint64_t*)nullptr, (const uint8_t*)nullptr, (uint64_t*)nullptr,
                          (unsigned long long*)nullptr, n, (hipStream_t)0);
    return bytes;
}

hipError_t select_u64(void* temp, size_t temp_bytes, const uint64_t* in, const uint8_t* flags, uint64_t* out,
                      unsigned long long* d_count, size_t n, hipStream_t stream)
{
    return rocprim::select(temp, temp_bytes, in, flags, out, d_count, n, stream);
}

}  // namespace fpx
