// zada_glsort.hip -- the one library call of the BZip2 path: a plain key-value radix sort (rocPRIM, ROCm's own primitives library)
// that puts the group lists of the rotation sort's late rounds into TEXT order (zada_bz2.hip, "Late rounds: group lists").  The
// entries are opaque 16-byte values here; nothing of the algorithm lives in this file.  (rocPRIM's headers want <cstring> first.)
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>

namespace zada {

// Sorts n (key, value) pairs by the key bits [begin_bit, end_bit), stable.  tmp == nullptr: only the size of the temporary storage
// is returned in tmp_bytes.  Returns 0 or the HIP error code.
struct Value16 { unsigned long long a, b; };
int gl_sort_pairs(hipStream_t st, void *tmp, size_t &tmp_bytes, const uint32_t *keys_in, uint32_t *keys_out,
                  const void *vals_in, void *vals_out, size_t n, unsigned begin_bit, unsigned end_bit) {
  return (int)rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, (const Value16 *)vals_in, (Value16 *)vals_out, n, begin_bit, end_bit, st);
}

}  // namespace zada
