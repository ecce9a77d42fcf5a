"""f3, the result record's wire format (include/orbfe_wire.h): host code, runs without a GPU.

Known answers are derived BY HAND from the reference's writer rules (src/WebSocket/bson.cpp:46-146,
bson.h:45-93) and its message fields (src/WebSocket/WebSocketCom.cpp:163-184); no bson module is
importable here, so the round trip uses the small decoder below (the same rules read backwards) and
the library's own orbfe_bson_find."""
import ctypes as C
import math
import os
import re
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    import orbfe
    if not os.path.exists(orbfe.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return orbfe.lib()


def decode(doc):
    """Minimal decoder of documents written by the reference's rules -> ordered list of (key, type, value)."""
    (total,) = struct.unpack_from("<I", doc, 0)
    assert total == len(doc) and doc[-1] == 0
    out, n = [], 4
    while n < total - 1:
        t = doc[n]
        n += 1
        end = doc.index(b"\x00", n)
        key = doc[n:end].decode()
        n = end + 1
        if t == 0x10:
            v = struct.unpack_from("<i", doc, n)[0]; n += 4
        elif t == 0x11:
            v = struct.unpack_from("<q", doc, n)[0]; n += 8
        elif t == 0x01:
            v = struct.unpack_from("<d", doc, n)[0]; n += 8
        elif t == 0x02:
            ln = struct.unpack_from("<I", doc, n)[0]; v = bytes(doc[n + 4:n + 4 + ln]); n += 4 + ln
        elif t == 0x05:
            ln = struct.unpack_from("<I", doc, n)[0]
            assert doc[n + 4] == 0x80, "the reference's binary subtype"
            v = bytes(doc[n + 5:n + 5 + ln]); n += 5 + ln
        else:
            raise AssertionError("type 0x%02x" % t)
        out.append((key, t, v))
    assert n == total - 1
    return out


def test_header_and_binding_agree(L):
    import orbfe
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "orbfe_wire.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(orbfe_(?:bson|wire)_[a-z0-9_]+)\s*\(", text)))
    assert declared == sorted(orbfe.WIRE_EXPORTS)
    for name in declared:
        assert hasattr(L, name)
    assert C.sizeof(orbfe.FrameMessage) == 64


def test_bson_writer_known_answer(L):
    """{"a": int32 7, "s": string "hi\\0" (3 bytes as passed), "b": binary 01 02} by the reference's rules:
    size = 4 + (1+2+4) + (1+2+4+3) + (1+2+4+1+2) + 1 = 32."""
    b = L.orbfe_bson_new()
    seven = C.c_int32(7)
    s = C.create_string_buffer(b"hi")      # 3 bytes with the terminator, which the CALLER includes
    raw = (C.c_uint8 * 2)(1, 2)
    assert L.orbfe_bson_add(b, b"a", 0x10, C.byref(seven), 0) == 0
    assert L.orbfe_bson_add(b, b"s", 0x02, s, 3) == 0
    assert L.orbfe_bson_add(b, b"b", 0x05, raw, 2) == 0
    assert L.orbfe_bson_process(b) == 0
    n = L.orbfe_bson_size(b)
    doc = C.string_at(L.orbfe_bson_ptr(b), n)
    want = bytes([32, 0, 0, 0,
                  0x10, ord("a"), 0, 7, 0, 0, 0,
                  0x02, ord("s"), 0, 3, 0, 0, 0, ord("h"), ord("i"), 0,
                  0x05, ord("b"), 0, 2, 0, 0, 0, 0x80, 1, 2,
                  0])
    assert doc == want
    assert decode(doc) == [("a", 0x10, 7), ("s", 0x02, b"hi\x00"), ("b", 0x05, b"\x01\x02")]
    assert L.orbfe_bson_add(b, b"late", 0x10, C.byref(seven), 0) != 0, "no add after process"
    L.orbfe_bson_free(b)
    # numbers: int64 and double sizes (bson.h:66-79)
    b = L.orbfe_bson_new()
    q, d = C.c_int64(-2), C.c_double(1.5)
    assert L.orbfe_bson_add(b, b"q", 0x11, C.byref(q), 0) == 0 and L.orbfe_bson_add(b, b"d", 0x01, C.byref(d), 0) == 0
    assert L.orbfe_bson_add(b, b"x", 0x07, C.byref(q), 0) != 0, "unknown type"
    L.orbfe_bson_process(b)
    doc = C.string_at(L.orbfe_bson_ptr(b), L.orbfe_bson_size(b))
    assert len(doc) == 4 + (1 + 2 + 8) * 2 + 1 and decode(doc) == [("q", 0x11, -2), ("d", 0x01, 1.5)]
    L.orbfe_bson_free(b)


@pytest.mark.parametrize("theta,want", [
    ((0.0, 0.0, math.pi / 2), (0, 0, 0)),
    ((0.5, -0.5, 0.0), (28, -29, -90)),          # 28.65 -> 28, -28.65 -> -29 (floor), -90
    ((math.pi, -math.pi, math.pi), (180, -181, 90)),  # float(pi) * 180 / pi = 180.00000500 -> 180; the negative floors to -181
])
def test_angles_as_websocketcom_computes_them(L, theta, want):
    out = (C.c_int32 * 3)()
    L.orbfe_wire_angles((C.c_float * 3)(*theta), out)
    # the reference: floor(float(theta.x) * 180 [float product] / pi [double]) etc.
    f = [np.float32(t) for t in theta]
    ref = (math.floor(float(np.float32(f[0] * np.float32(180))) / math.pi),
           math.floor(float(np.float32(f[1] * np.float32(180))) / math.pi),
           math.floor((float(f[2]) - math.pi / 2) * 180 / math.pi))
    assert tuple(out) == ref == want


def test_frame_message_round_trip_and_layout(L):
    import orbfe
    rng = np.random.default_rng(11)
    n = 37
    kx = rng.integers(0, 848, n).astype(np.uint16)
    ky = rng.integers(0, 480, n).astype(np.uint16)
    img = rng.integers(0, 256, 1001).astype(np.uint8)
    m = orbfe.FrameMessage((C.c_float * 3)(0.1, -0.2, 1.7), 848, 480, 1, kx.ctypes.data, ky.ctypes.data, n, img.ctypes.data,
                           img.size)
    need = L.orbfe_wire_frame_size(C.byref(m))
    # 4 + six int32 fields (1 + key + 1 + 4) + three binaries (1 + key + 1 + 4 + 1 + bytes) + 1
    keys6 = ["ax", "ay", "az", "width", "height", "channels"]
    expect = 4 + sum(1 + len(k) + 1 + 4 for k in keys6) + (1 + 11 + 1 + 5 + 2 * n) * 2 + (1 + 5 + 1 + 5 + img.size) + 1
    assert need == expect
    buf = (C.c_uint8 * need)()
    wr = C.c_size_t()
    assert L.orbfe_wire_frame_encode(C.byref(m), buf, need - 1, C.byref(wr)) == orbfe.ERR_CAPACITY and wr.value == need
    assert L.orbfe_wire_frame_encode(C.byref(m), buf, need, C.byref(wr)) == 0
    doc = bytes(buf)
    fields = decode(doc)
    assert [k for k, _, _ in fields] == keys6 + ["keypoints_x", "keypoints_y", "image"]  # WebSocketCom.cpp:168-184 order
    d = {k: v for k, _, v in fields}
    ang = (C.c_int32 * 3)()
    L.orbfe_wire_angles(m.theta, ang)
    assert (d["ax"], d["ay"], d["az"]) == tuple(ang) == (5, -12, 7)
    assert (d["width"], d["height"], d["channels"]) == (848, 480, 1)
    assert d["keypoints_x"] == kx.tobytes() and d["keypoints_y"] == ky.tobytes() and d["image"] == img.tobytes()
    # the same document through the generic writer, as WebSocketCom builds it
    b = L.orbfe_bson_new()
    vals = [C.c_int32(v) for v in (d["ax"], d["ay"], d["az"], 848, 480, 1)]
    for k, v in zip(keys6, vals):
        assert L.orbfe_bson_add(b, k.encode(), 0x10, C.byref(v), 0) == 0
    assert L.orbfe_bson_add(b, b"keypoints_x", 0x05, kx.ctypes.data, 2 * n) == 0
    assert L.orbfe_bson_add(b, b"keypoints_y", 0x05, ky.ctypes.data, 2 * n) == 0
    assert L.orbfe_bson_add(b, b"image", 0x05, img.ctypes.data, img.size) == 0
    L.orbfe_bson_process(b)
    assert C.string_at(L.orbfe_bson_ptr(b), L.orbfe_bson_size(b)) == doc
    L.orbfe_bson_free(b)
    # the library's own reader
    val, nb = C.c_void_p(), C.c_size_t()
    assert L.orbfe_bson_find(buf, need, b"keypoints_y", C.byref(val), C.byref(nb)) == 0x05 and nb.value == 2 * n
    assert C.string_at(val.value, nb.value) == ky.tobytes()
    assert L.orbfe_bson_find(buf, need, b"height", C.byref(val), C.byref(nb)) == 0x10
    assert struct.unpack("<i", C.string_at(val.value, 4))[0] == 480
    assert L.orbfe_bson_find(buf, need, b"nope", None, None) == -1
    assert L.orbfe_bson_find(buf, need - 3, b"image", None, None) == -1, "truncated document"


def test_empty_frame_message(L):
    import orbfe
    m = orbfe.FrameMessage((C.c_float * 3)(0, 0, 0), 640, 480, 1, None, None, 0, None, 0)
    need = L.orbfe_wire_frame_size(C.byref(m))
    buf = (C.c_uint8 * need)()
    assert L.orbfe_wire_frame_encode(C.byref(m), buf, need, None) == 0
    d = {k: v for k, _, v in decode(bytes(buf))}
    assert d["keypoints_x"] == b"" and d["image"] == b"" and d["az"] == -90


def test_frame_message_through_an_independent_bson_decoder(L):
    """The viewer decodes the message with the npm `bson` package's BSON.deserialize and reads msg.width / height /
    channels, msg.image.buffer, msg.keypoints_x.buffer, msg.keypoints_y.buffer, msg.ax / ay / az
    (CarDriver/src/hooks/useWebsockets.js:30-71; package.json: bson ^4.2.2 -- not vendored, cannot be run here).
    PyMongo's `bson` is the same specification implemented by the same vendor and IS in this image: the bytes of
    orbfe_wire_frame_encode must decode with it into exactly those fields -- int32 scalars and generic binaries
    (subtype 0x80, user-defined, as bson.h:18 has it: the viewer only touches `.buffer`), in the producer's order
    (WebSocketCom.cpp:167-184)."""
    bson = pytest.importorskip("bson")
    if not hasattr(bson, "decode"):
        pytest.skip("a `bson` module without decode(): not PyMongo's")
    import orbfe
    rng = np.random.default_rng(12)
    for n, img_bytes in ((0, 0), (1, 7), (405, 848 * 480)):
        kx = rng.integers(0, 848, max(n, 1)).astype(np.uint16)[:n]
        ky = rng.integers(0, 480, max(n, 1)).astype(np.uint16)[:n]
        img = rng.integers(0, 256, max(img_bytes, 1)).astype(np.uint8)[:img_bytes]
        m = orbfe.FrameMessage((C.c_float * 3)(0.3, -1.1, 2.0), 848, 480, 1, kx.ctypes.data if n else None,
                               ky.ctypes.data if n else None, n, img.ctypes.data if img_bytes else None, img_bytes)
        need = L.orbfe_wire_frame_size(C.byref(m))
        buf = (C.c_uint8 * need)()
        assert L.orbfe_wire_frame_encode(C.byref(m), buf, need, None) == 0
        msg = bson.decode(bytes(buf))  # strict: raises on a malformed document, a bad length or trailing bytes
        assert list(msg) == ["ax", "ay", "az", "width", "height", "channels", "keypoints_x", "keypoints_y", "image"]
        ang = (C.c_int32 * 3)()
        L.orbfe_wire_angles(m.theta, ang)
        assert (msg["ax"], msg["ay"], msg["az"]) == tuple(ang)
        assert all(type(msg[k]) is int for k in ("ax", "ay", "az", "width", "height", "channels"))
        assert (msg["width"], msg["height"], msg["channels"]) == (848, 480, 1)
        # what the viewer wraps: new Uint8Array(msg.image.buffer), Uint8Array.from(msg.keypoints_x.buffer)
        assert bytes(msg["keypoints_x"]) == kx.tobytes() and bytes(msg["keypoints_y"]) == ky.tobytes()
        assert bytes(msg["image"]) == img.tobytes()
        for k in ("keypoints_x", "keypoints_y", "image"):
            assert msg[k].subtype == 0x80, "the user-defined binary subtype the reference writes (bson.h:18)"
        # and the other way round: the library's reader on a document PyMongo wrote
        theirs = bson.encode({"width": 848, "image": bson.Binary(img.tobytes(), 0), "height": 480})
        raw = (C.c_uint8 * len(theirs)).from_buffer_copy(theirs)
        val, nb = C.c_void_p(), C.c_size_t()
        assert L.orbfe_bson_find(raw, len(theirs), b"image", C.byref(val), C.byref(nb)) == 0x05 and nb.value == img_bytes
        assert C.string_at(val.value, nb.value) == img.tobytes()
        assert L.orbfe_bson_find(raw, len(theirs), b"height", C.byref(val), C.byref(nb)) == 0x10
