// wire_bson.cpp -- include/orbfe_wire.h: the viewer message of the reference's WebSocketCom thread
// (src/WebSocket/WebSocketCom.cpp:163-184) as the document its own writer produces
// (src/WebSocket/bson.h:39-107, bson.cpp:46-146).  Host code only.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/orbfe.h"
#include "../../include/orbfe_wire.h"

namespace {

struct Item {
    std::string key;
    int type;
    const void *value;
    uint32_t bytes;
};

// value part of an element: what Bson::add adds to size_ beyond type byte and key (bson.h:66-88)
size_t value_span(int type, size_t bytes)
{
    switch (type) {
    case ORBFE_BSON_DOUBLE: return 8;
    case ORBFE_BSON_INT32: return 4;
    case ORBFE_BSON_INT64: return 8;
    case ORBFE_BSON_STRING: return bytes + 4;
    case ORBFE_BSON_BINARY: return bytes + 1 + 4; // + subtype + length
    default: return 0;
    }
}

void put_u32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); } // the reference stores through uint32_t* (little-endian hosts)

// one element, bson.cpp:62-118; returns the bytes written
size_t put_item(uint8_t *p, const Item &it)
{
    size_t n = 0;
    p[n++] = (uint8_t)it.type;
    memcpy(p + n, it.key.c_str(), it.key.size() + 1);
    n += it.key.size() + 1;
    switch (it.type) {
    case ORBFE_BSON_DOUBLE:
    case ORBFE_BSON_INT32:
    case ORBFE_BSON_INT64:
        memcpy(p + n, it.value, it.bytes);
        n += it.bytes;
        break;
    case ORBFE_BSON_STRING:
        put_u32(p + n, it.bytes);
        n += 4;
        if (it.bytes) memcpy(p + n, it.value, it.bytes);
        n += it.bytes;
        break;
    case ORBFE_BSON_BINARY:
        put_u32(p + n, it.bytes);
        n += 4;
        p[n++] = (uint8_t)ORBFE_BSON_BINARY_SUBTYPE;
        if (it.bytes) memcpy(p + n, it.value, it.bytes);
        n += it.bytes;
        break;
    default: break;
    }
    return n;
}

} // namespace

struct orbfe_bson {
    std::vector<Item> items;
    uint32_t size = 4; // uint32_t size_ = sizeof(uint32_t), bson.h:101
    std::vector<uint8_t> buffer;
    bool processed = false;
};

extern "C" {

orbfe_bson *orbfe_bson_new(void) { return new orbfe_bson(); }
void orbfe_bson_free(orbfe_bson *b) { delete b; }

int orbfe_bson_add(orbfe_bson *b, const char *key, int type, const void *value, size_t value_bytes)
{
    if (!b || !key || b->processed) return ORBFE_ERR_INVALID_ARG;
    if (type != ORBFE_BSON_DOUBLE && type != ORBFE_BSON_INT32 && type != ORBFE_BSON_INT64 && type != ORBFE_BSON_STRING &&
        type != ORBFE_BSON_BINARY)
        return ORBFE_ERR_INVALID_ARG;
    const bool sized = type == ORBFE_BSON_STRING || type == ORBFE_BSON_BINARY;
    if ((!value && !(sized && value_bytes == 0)) || value_bytes > 0x7FFFFFFFu) return ORBFE_ERR_INVALID_ARG;
    Item it;
    it.key = key;
    it.type = type;
    it.value = value;
    it.bytes = sized ? (uint32_t)value_bytes : (type == ORBFE_BSON_INT32 ? 4u : 8u);
    b->size += 1 + (uint32_t)it.key.size() + 1 + (uint32_t)value_span(type, value_bytes);
    b->items.push_back(it);
    return ORBFE_OK;
}

int orbfe_bson_process(orbfe_bson *b)
{
    if (!b || b->processed) return ORBFE_ERR_INVALID_ARG;
    b->size += 1; // trailing 0x00, bson.cpp:48
    b->buffer.assign(b->size, 0);
    uint8_t *p = b->buffer.data();
    put_u32(p, b->size);
    size_t n = 4;
    for (const Item &it : b->items) n += put_item(p + n, it);
    p[n] = 0;
    b->processed = true;
    return ORBFE_OK;
}

const uint8_t *orbfe_bson_ptr(const orbfe_bson *b) { return b && b->processed ? b->buffer.data() : nullptr; }
uint32_t orbfe_bson_size(const orbfe_bson *b) { return b ? b->size : 0; }

void orbfe_wire_angles(const float theta[3], int32_t out[3])
{
    const double pi = 3.14159265358979323846; // CUDART_PI_D
    // WebSocketCom.cpp:165-167: `theta.x * 180` is a float product (float * int), the division is double
    out[0] = (int32_t)floor((double)(theta[0] * 180) / pi);
    out[1] = (int32_t)floor((double)(theta[1] * 180) / pi);
    out[2] = (int32_t)floor(((double)theta[2] - pi / 2) * 180 / pi);
}

static void frame_items(const orbfe_frame_message *m, int32_t scalars[6], Item items[9])
{
    orbfe_wire_angles(m->theta, scalars);
    scalars[3] = m->width;
    scalars[4] = m->height;
    scalars[5] = m->channels;
    static const char names[9][12] = {"ax", "ay", "az", "width", "height", "channels", "keypoints_x", "keypoints_y", "image"};
    for (int i = 0; i < 6; i++) items[i] = Item{names[i], ORBFE_BSON_INT32, &scalars[i], 4u};
    const uint32_t kb = (uint32_t)(m->matched_keypoints > 0 ? m->matched_keypoints : 0) * (uint32_t)sizeof(uint16_t);
    items[6] = Item{names[6], ORBFE_BSON_BINARY, m->keypoints_x, kb};
    items[7] = Item{names[7], ORBFE_BSON_BINARY, m->keypoints_y, kb};
    items[8] = Item{names[8], ORBFE_BSON_BINARY, m->image, (uint32_t)m->image_length};
}

size_t orbfe_wire_frame_size(const orbfe_frame_message *m)
{
    if (!m) return 0;
    int32_t sc[6];
    Item items[9];
    frame_items(m, sc, items);
    size_t n = 4 + 1;
    for (const Item &it : items) n += 1 + it.key.size() + 1 + value_span(it.type, it.bytes);
    return n;
}

int orbfe_wire_frame_encode(const orbfe_frame_message *m, uint8_t *out, size_t cap, size_t *written)
{
    if (!m || !out || m->image_length > 0x7FFFFFFFu || (m->matched_keypoints > 0 && (!m->keypoints_x || !m->keypoints_y)) ||
        (m->image_length > 0 && !m->image))
        return ORBFE_ERR_INVALID_ARG;
    const size_t need = orbfe_wire_frame_size(m);
    if (written) *written = need;
    if (cap < need) return ORBFE_ERR_CAPACITY;
    int32_t sc[6];
    Item items[9];
    frame_items(m, sc, items);
    put_u32(out, (uint32_t)need);
    size_t n = 4;
    for (const Item &it : items) n += put_item(out + n, it);
    out[n] = 0;
    return ORBFE_OK;
}

int orbfe_bson_find(const uint8_t *doc, size_t doc_bytes, const char *key, const uint8_t **value, size_t *value_bytes)
{
    if (!doc || !key || doc_bytes < 5) return -1;
    uint32_t total;
    memcpy(&total, doc, 4);
    if (total > doc_bytes || total < 5 || doc[total - 1] != 0) return -1;
    size_t n = 4;
    while (n < total - 1) {
        const int type = doc[n++];
        const size_t k0 = n;
        while (n < total - 1 && doc[n] != 0) n++;
        if (n >= total - 1) return -1;
        const size_t klen = n - k0;
        n++; // the key's terminator
        size_t payload = 0, skip = 0;
        uint32_t len = 0;
        switch (type) {
        case ORBFE_BSON_DOUBLE:
        case ORBFE_BSON_INT64: payload = 8; break;
        case ORBFE_BSON_INT32: payload = 4; break;
        case ORBFE_BSON_STRING:
            if (n + 4 > total) return -1;
            memcpy(&len, doc + n, 4);
            skip = 4;
            payload = len;
            break;
        case ORBFE_BSON_BINARY:
            if (n + 5 > total) return -1;
            memcpy(&len, doc + n, 4);
            skip = 5;
            payload = len;
            break;
        default: return -1;
        }
        if (n + skip + payload > total - 1) return -1;
        if (klen == strlen(key) && memcmp(doc + k0, key, klen) == 0) {
            if (value) *value = doc + n + skip;
            if (value_bytes) *value_bytes = payload;
            return type;
        }
        n += skip + payload;
    }
    return -1;
}

} // extern "C"
