/* A C99 consumer of the C ABI (include/densereg.h): what a maintainer binding the library from another language links
 * against.  Calls only the host-side entry points, so it runs without a GPU:
 *
 *   gcc -std=c99 -Iinclude examples/abi_probe.c -o /tmp/abi_probe -Ldensereg_amd/lib -ldensereg_hip \
 *       -Wl,-rpath,$PWD/densereg_amd/lib -Wl,-rpath,/opt/rocm/lib
 *   /tmp/abi_probe
 */
#include <stdio.h>
#include <string.h>

#include "densereg.h"

int main(void) {
    /* CRC-32C check value of RFC 3720 B.4 */
    const uint32_t crc = dr_crc32c(0, "123456789", 9);
    /* two rows of a 2-byte-per-pixel image: filter type 1 (Sub), then 2 (Up) */
    const uint8_t filtered[2 * 5] = {1, 10, 20, 1, 2, 2, 5, 5, 5, 5};
    uint8_t out[8];
    const int rc = dr_png_unfilter(filtered, 2, 4, 2, out);
    const uint8_t expect[8] = {10, 20, 11, 22, 15, 25, 16, 27};
    printf("abi=%d backend=%s crc32c=%08x unfilter_rc=%d unfilter_ok=%d\n", dr_abi_version(), dr_backend(), crc, rc,
           memcmp(out, expect, 8) == 0);
    return (crc == 0xE3069283u && rc == DR_OK && memcmp(out, expect, 8) == 0) ? 0 : 1;
}
