"""A second, independent statement of the MCAP record layouts (https://mcap.dev/spec) for the tests of
cloudini_amd/csrc/host/mcap_io.cpp: a writer of uncompressed files and a reader that walks every record, the summary section
included. Test infrastructure only."""
import struct

MAGIC = b"\x89MCAP0\r\n"
(HEADER, FOOTER, SCHEMA, CHANNEL, MESSAGE, CHUNK, MESSAGE_INDEX, CHUNK_INDEX, ATTACHMENT, ATTACHMENT_INDEX, STATISTICS, METADATA,
 METADATA_INDEX, SUMMARY_OFFSET, DATA_END) = range(1, 16)


def _s(x: str) -> bytes:
    b = x.encode()
    return struct.pack("<I", len(b)) + b


def _map(m) -> bytes:
    body = b"".join(_s(k) + _s(v) for k, v in m)
    return struct.pack("<I", len(body)) + body


def _rec(op: int, body: bytes) -> bytes:
    return struct.pack("<BQ", op, len(body)) + body


def write(path, profile, schemas, channels, messages, metadata=(), chunk_messages=None, declare_late=False):
    """schemas: [(id, name, encoding, data)], channels: [(id, schema_id, topic, encoding, [(k, v)])],
    messages: [(channel, seq, log, pub, bytes)]; chunk_messages: messages per (uncompressed) chunk, None = no chunks.
    declare_late: a Schema / Channel record stands right in front of the first message that needs it (wherever in the
    file that is) instead of at the file's beginning."""
    out = [MAGIC, _rec(HEADER, _s(profile) + _s("mcap_py"))]
    srec = {i: _rec(SCHEMA, struct.pack("<H", i) + _s(n) + _s(e) + struct.pack("<I", len(d)) + d) for i, n, e, d in schemas}
    crec = {i: (s, _rec(CHANNEL, struct.pack("<HH", i, s) + _s(t) + _s(e) + _map(m))) for i, s, t, e, m in channels}
    decl = b"" if declare_late else b"".join(srec.values()) + b"".join(r for _, r in crec.values())
    msgs, seen_s, seen_c = [], set(), set()
    for c, q, lt, pt, d in messages:
        pre = b""
        if declare_late and c not in seen_c:
            sid = crec[c][0]
            if sid not in seen_s and sid in srec:
                pre += srec[sid]
                seen_s.add(sid)
            pre += crec[c][1]
            seen_c.add(c)
        msgs.append(pre + _rec(MESSAGE, struct.pack("<HIQQ", c, q, lt, pt) + d))
    if chunk_messages is None:
        out.append(decl)
        out.extend(msgs)
    else:
        first = True
        for k in range(0, max(1, len(msgs)), chunk_messages):
            body = (decl if first else b"") + b"".join(msgs[k:k + chunk_messages])
            first = False
            times = [m[2] for m in messages[k:k + chunk_messages]] or [0]
            out.append(_rec(CHUNK, struct.pack("<QQQI", min(times), max(times), len(body), 0) + _s("") + struct.pack("<Q", len(body)) + body))
    for name, entries in metadata:
        out.append(_rec(METADATA, _s(name) + _map(entries)))
    out.append(_rec(DATA_END, struct.pack("<I", 0)))
    out.append(_rec(FOOTER, struct.pack("<QQI", 0, 0, 0)))
    out.append(MAGIC)
    with open(path, "wb") as f:
        f.write(b"".join(out))


class _Cur:
    def __init__(self, b):
        self.b, self.i = b, 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.i)
        self.i += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def s(self):
        n = self.take("I")
        v = self.b[self.i:self.i + n].decode()
        assert len(v.encode()) == n
        self.i += n
        return v

    def m(self):
        n = self.take("I")
        end = self.i + n
        out = []
        while self.i < end:
            out.append((self.s(), self.s()))
        assert self.i == end
        return out

    def rest(self):
        v = self.b[self.i:]
        self.i = len(self.b)
        return v


def read(path):
    """Every record of an uncompressed file -> dict(header, schemas, channels, messages, metadata, chunks, summary)."""
    b = open(path, "rb").read()
    assert b[:8] == MAGIC and b[-8:] == MAGIC
    r = {"schemas": {}, "channels": {}, "messages": [], "metadata": [], "chunks": [], "summary": {}, "records": []}

    def walk(buf, base, section):
        i = 0
        while i < len(buf):
            op, n = struct.unpack_from("<BQ", buf, i)
            body = buf[i + 9:i + 9 + n]
            assert len(body) == n, "record longer than the file"
            off = base + i
            i += 9 + n
            c = _Cur(body)
            r["records"].append((op, off, 9 + n, section))
            if op == HEADER:
                r["header"] = (c.s(), c.s())
            elif op == SCHEMA:
                sid, name, enc = c.take("H"), c.s(), c.s()
                d = c.rest()
                (dn,) = struct.unpack_from("<I", d, 0)
                assert dn == len(d) - 4
                (r["summary"].setdefault("schemas", {}) if section == "summary" else r["schemas"])[sid] = (name, enc, d[4:])
            elif op == CHANNEL:
                cid, sid, topic, enc, md = c.take("H"), c.take("H"), c.s(), c.s(), c.m()
                (r["summary"].setdefault("channels", {}) if section == "summary" else r["channels"])[cid] = (sid, topic, enc, md)
            elif op == MESSAGE:
                ch, seq, lt, pt = c.take("HIQQ")
                r["messages"].append((ch, seq, lt, pt, c.rest()))
            elif op == CHUNK:
                t0, t1, usize, crc = c.take("QQQI")
                comp = c.s()
                n2 = c.take("Q")
                recs = c.rest()
                assert len(recs) == n2
                r["chunks"].append((off, 9 + n, t0, t1, usize, comp, n2))
                assert comp == "", "the Python reader takes uncompressed chunks only"
                assert usize == n2
                walk(recs, off + 9 + (len(body) - n2), "chunk")
            elif op == MESSAGE_INDEX:
                ch = c.take("H")
                nb = c.take("I")
                assert nb % 16 == 0 and nb == len(body) - 6
                r.setdefault("message_index", []).append((off, 9 + n, ch, [c.take("QQ") for _ in range(nb // 16)]))
            elif op == METADATA:
                r["metadata"].append((c.s(), c.m()))
            elif op == DATA_END:
                assert c.take("I") == 0
                section = "summary"
            elif op == CHUNK_INDEX:
                t0, t1, coff, clen = c.take("QQQQ")
                mio = c.take("I")
                assert mio % 10 == 0
                offs = dict(c.take("HQ") for _ in range(mio // 10))
                mil = c.take("Q")
                comp = c.s()
                csize, usize = c.take("QQ")
                r["summary"].setdefault("chunk_index", []).append((t0, t1, coff, clen, mil, comp, csize, usize))
                r["summary"].setdefault("chunk_index_message_offsets", []).append(offs)
            elif op == STATISTICS:
                mc, sc, cc, ac, mdc, chc, t0, t1 = c.take("QHIIIIQQ")
                n3 = c.take("I")
                counts = {}
                for _ in range(n3 // 10):
                    k, v = c.take("HQ")
                    counts[k] = v
                r["summary"]["statistics"] = dict(messages=mc, schemas=sc, channels=cc, attachments=ac, metadata=mdc, chunks=chc,
                                                  t0=t0, t1=t1, counts=counts)
            elif op == SUMMARY_OFFSET:
                g, st, ln = c.take("BQQ")
                r["summary"].setdefault("offsets", []).append((g, st, ln))
            elif op == FOOTER:
                r["footer"] = c.take("QQI")
                assert i == len(buf), "records behind the footer"
        return

    walk(b[8:-8], 8, "data")
    return r
