// bfk_sort.hip — the one library primitive of the prefix-group path: a device radix sort of (32-bit key, row) records
// (rocPRIM, header-only).  In a translation unit of its own: the rocPRIM templates take longer to compile than all
// of bfk_kernels.hip.
#include <cstdint>
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

namespace bfk {

// temp == nullptr: only *temp_bytes is set.  Sorts by key bits [0, bits); keys_in / rows_in stay intact.
int sort_records(void *temp, size_t *temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const int *rows_in,
                 int *rows_out, size_t n, int bits, hipStream_t st) {
    // (the library's default hands inputs of up to 1M items to a merge sort, which does not profit from the few key bits:
    // 143 us for the 700k records of 100k rows, ten launches each of two kernels; the radix passes take over from 64k items)
    using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 65536>;
    return (int)rocprim::radix_sort_pairs<cfg>(temp, *temp_bytes, keys_in, keys_out, rows_in, rows_out, n, 0u, (unsigned)bits, st);
}

}  // namespace bfk
