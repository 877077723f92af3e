// Host-side CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slicing-by-8.  Used by tf_checkpoint.py for the block
// trailers of the TensorBundle index table and the per-tensor checksums of BundleEntryProto (the format tf.train.Checkpoint
// writes, reference UNet/train.py:96,184; UNet/model.py:81-83): 124 MB of weights + two Adam slots per checkpoint is too much
// for a byte loop in Python.  Pure host code; no device work.
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace {
struct Tables {
    uint32_t t[8][256];
    Tables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFFu];
    }
};
const Tables& tables() { static const Tables T; return T; }
}  // namespace

// crc32c::Extend(init, data, n) of TensorFlow's lib/hash/crc32c.h: `init` is the CRC of the bytes before `data` (0 to start).
extern "C" uint32_t unet_crc32c_extend(uint32_t init, const void* data, size_t n) {
    const Tables& T = tables();
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = ~init;
    while (n && (reinterpret_cast<uintptr_t>(p) & 7u)) { c = T.t[0][(c ^ *p++) & 0xFFu] ^ (c >> 8); --n; }
    while (n >= 8) {
        uint64_t w; std::memcpy(&w, p, 8);
        w ^= c;
        c = T.t[7][w & 0xFF] ^ T.t[6][(w >> 8) & 0xFF] ^ T.t[5][(w >> 16) & 0xFF] ^ T.t[4][(w >> 24) & 0xFF] ^
            T.t[3][(w >> 32) & 0xFF] ^ T.t[2][(w >> 40) & 0xFF] ^ T.t[1][(w >> 48) & 0xFF] ^ T.t[0][(w >> 56) & 0xFF];
        p += 8; n -= 8;
    }
    while (n--) c = T.t[0][(c ^ *p++) & 0xFFu] ^ (c >> 8);
    return ~c;
}
