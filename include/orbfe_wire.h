/* orbfe_wire.h -- the result record on the wire: the BSON message the reference's
 * WebSocketCom thread sends to the viewer for every slam frame
 * (src/WebSocket/WebSocketCom.cpp:167-184), written by the rules of its own writer
 * (src/WebSocket/bson.h:39-107, bson.cpp:46-146).  Pure host code (no HIP), part of liborbfe.so,
 * so that the new front end can feed the existing viewer (CarDriver/src/hooks/useWebsockets.js:30-71)
 * unchanged.
 *
 * Document layout as the reference writes it (little endian):
 *     int32 total_size | element* | 0x00
 *     element = type byte | key bytes | 0x00 | value
 *     int32 (0x10) / int64 (0x11) / double (0x01): the raw value
 *     string (0x02): int32 n | n bytes  -- n is what the caller passed: the reference's writer does
 *                    NOT append the terminating 0x00 standard BSON strings carry (bson.cpp:92-100)
 *     binary (0x05): int32 n | subtype 0x80 | n bytes
 */
#ifndef ORBFE_WIRE_H
#define ORBFE_WIRE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enum class bson_value_type, src/WebSocket/bson.h:11-19 */
enum {
    ORBFE_BSON_DOUBLE = 0x01,
    ORBFE_BSON_STRING = 0x02,
    ORBFE_BSON_BINARY = 0x05,
    ORBFE_BSON_INT32 = 0x10,
    ORBFE_BSON_INT64 = 0x11,
    ORBFE_BSON_BINARY_SUBTYPE = 0x80
};

/* ---- the reference's Bson class (add ... add, process, ptr, size) as a C API ---------------- */
typedef struct orbfe_bson orbfe_bson;
orbfe_bson *orbfe_bson_new(void);
void orbfe_bson_free(orbfe_bson *b);
/* Bson::add, bson.h:45-93.  value_bytes is used for strings and binaries only (numbers have their
 * own size).  The value is NOT copied: like the reference, the pointer must stay valid until
 * orbfe_bson_process.  Returns ORBFE_OK or ORBFE_ERR_INVALID_ARG. */
int orbfe_bson_add(orbfe_bson *b, const char *key, int value_type, const void *value, size_t value_bytes);
/* Bson::process, bson.cpp:46-131: builds the document; afterwards ptr / size give it. */
int orbfe_bson_process(orbfe_bson *b);
const uint8_t *orbfe_bson_ptr(const orbfe_bson *b);
uint32_t orbfe_bson_size(const orbfe_bson *b);

/* ---- the frame message, WebSocketCom.cpp:163-184 ------------------------------------------- */
typedef struct orbfe_frame_message {
    float theta[3];              /* slam_frame_t::theta (IMU complementary filter), radians          */
    int32_t width, height;       /* _ctx->cam_w, cam_h                                                */
    int32_t channels;            /* 1                                                                 */
    const uint16_t *keypoints_x; /* matched current keypoints (orbfe_match_compact)                   */
    const uint16_t *keypoints_y;
    int32_t matched_keypoints;   /* slam_frame_t::h_matched_keypoints_num                             */
    const uint8_t *image;        /* the preview image bytes (nvJPEG output in the reference)          */
    size_t image_length;
} orbfe_frame_message;

/* ax, ay, az as WebSocketCom.cpp:165-167 computes them: floor(theta.x * 180 / pi),
 * floor(theta.y * 180 / pi), floor((theta.z - pi / 2) * 180 / pi) with the reference's operand
 * types (float * int, then / double). */
void orbfe_wire_angles(const float theta[3], int32_t out[3]);

/* Size of the encoded message, and the encoder: fields ax, ay, az, width, height, channels (int32),
 * keypoints_x, keypoints_y, image (binary), in that order.  Returns ORBFE_ERR_CAPACITY if cap is too small. */
size_t orbfe_wire_frame_size(const orbfe_frame_message *m);
int orbfe_wire_frame_encode(const orbfe_frame_message *m, uint8_t *out, size_t cap, size_t *written);

/* Decoder for clients / round-trip tests: finds element `key` in a document written by the rules above;
 * *value points INTO doc (for binaries past the subtype byte), *value_bytes is its payload size.
 * Returns the element's type byte, or -1 if the key is absent or the document malformed. */
int orbfe_bson_find(const uint8_t *doc, size_t doc_bytes, const char *key, const uint8_t **value, size_t *value_bytes);

#ifdef __cplusplus
}
#endif
#endif /* ORBFE_WIRE_H */
