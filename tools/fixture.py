"""Fixture / wire format for pinning the engine against a run of the Go reference.

One file = one recorded call of the hot path (inputs AND the reference's output), little-endian:

    bytes 0..7    magic  b"MKHEFIX1"
    bytes 8..11   uint32 header length H
    bytes 12..    H bytes of UTF-8 JSON: {"meta": {...}, "arrays": [{"name", "shape", "offset"}, ...]}
    then          zero padding to a multiple of 8, then the payload: uint64 words; `offset` counts words from
                  the start of the payload

`meta` of a "mkrlwe.MulAndRelin" record: op, logN, Q, P, gamma, psiQ, psiP (plain 2N-th roots lattigo used:
InvMForm(ring.NttPsi[i][N/2])), level, ids0, ids1 (party id strings).  Arrays (SwitchingKey = [beta][nQ+nP][N],
polynomial = [level+1][N], exactly the words of the Go slices): "crs_u", "rlk/<id>/b|d|v", "op0/<id>", "op1/<id>"
("0" for the c_0 component), "out/<id>".  Writer on the Go side: shim/go/dump/dump_test.go.
"""
import json
import struct

import numpy as np

MAGIC = b"MKHEFIX1"


def write(path, meta, arrays):
    """arrays: {name: uint64 ndarray}"""
    entries, off, blobs = [], 0, []
    for name, a in arrays.items():
        a = np.ascontiguousarray(a, dtype="<u8")
        entries.append(dict(name=name, shape=list(a.shape), offset=off))
        off += a.size
        blobs.append(a)
    header = json.dumps(dict(meta=meta, arrays=entries)).encode()
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<I", len(header)))
        f.write(header)
        f.write(b"\0" * ((-(12 + len(header))) % 8))
        for a in blobs:
            f.write(a.tobytes())


def read(path):
    """-> (meta, {name: uint64 ndarray})"""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:8] != MAGIC:
        raise ValueError("not an MKHEFIX1 fixture")
    (hlen,) = struct.unpack("<I", raw[8:12])
    hdr = json.loads(raw[12:12 + hlen].decode())
    start = 12 + hlen
    start += (-start) % 8
    payload = np.frombuffer(raw, dtype="<u8", offset=start)
    arrays = {}
    for e in hdr["arrays"]:
        n = int(np.prod(e["shape"])) if e["shape"] else 1
        if e["offset"] + n > payload.size:
            raise ValueError("fixture truncated at array %s" % e["name"])
        arrays[e["name"]] = payload[e["offset"]:e["offset"] + n].reshape(e["shape"]).astype(np.uint64)
    return hdr["meta"], arrays
