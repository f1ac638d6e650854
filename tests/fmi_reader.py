"""Independent pure-Python reader of the reference's .fmi files (Boost binary archive, layout in
SURVEY.md §8b).  Test-side only: cross-checks the product's C++ reader and feeds the oracle."""
import struct


class _R:
    def __init__(self, d):
        self.d, self.p = d, 0

    def u(self, fmt):
        v = struct.unpack_from("<" + fmt, self.d, self.p)
        self.p += struct.calcsize("<" + fmt)
        return v[0] if len(v) == 1 else v

    def s(self):
        n = self.u("Q")
        v = self.d[self.p:self.p + n]
        self.p += n
        return v.decode("utf-8")

    def ci(self):
        self.u("B")
        return self.u("I")

    def vec(self, fmt):
        n = self.u("Q")
        return [self.u(fmt) for _ in range(n)]


def read_fmi(path):
    r = _R(open(path, "rb").read())
    assert r.d[:3] == b"FMI"
    r.p = 4
    assert r.s() == "serialization::archive"
    lvt = r.u("H")
    assert r.d[r.p:r.p + 4] == bytes([4, 8, 4, 8])
    r.p += 4
    assert r.u("I") == 1
    out = {"lvt": lvt}
    r.ci()
    out["pt"] = r.u("i")
    r.u("h")
    sai_version = r.ci()
    r.u("I")
    r.ci()
    r.ci()
    n = r.u("Q")
    r.u("I")
    out["forms"] = [r.s() for _ in range(n)]
    out["sfreq"] = r.vec("I")
    r.ci()
    map_start = r.p
    n = r.u("Q")
    r.u("Q")
    r.u("I")
    f2i = {}
    for i in range(n):
        if i == 0:
            r.ci()
        k = r.s()
        f2i[k] = r.u("I")
    out["form2index"] = f2i
    out["map_span"] = (map_start, r.p)  # the unordered_map's count, bucket count and items: order is the writer's C++ library's
    sa_version = r.ci()
    out["sa_version"] = sa_version
    out["sorted"] = r.u("B")
    r.ci()
    n = r.u("Q")
    suf = []
    if sa_version == 1:
        r.u("I")
        if n:
            r.ci()
        for _ in range(n):
            suf.append((r.u("I"), r.u("H")))
    else:
        for _ in range(n):
            sid, pos, _pad = r.u("IHH")
            suf.append((sid, pos))
    out["suffixes"] = suf
    out["buffer"] = r.vec("I")
    out["sent_pos"] = r.vec("I")
    out["qva"] = r.vec("I")
    n = r.u("Q")
    r.u("I")
    out["ids"] = [r.s() for _ in range(n)]
    r.ci()
    n = r.u("Q")
    r.u("I")
    real = []
    map_seen = pair_seen = False
    bucket_fields = []  # offsets of the per-sentence unordered_map bucket counts (the writer's C++ library's choice)
    for i in range(n):
        if i == 0:
            r.ci()
        ts = r.s()
        if not map_seen:
            r.ci()
            map_seen = True
        m = r.u("Q")
        bucket_fields.append(r.p)
        r.u("Q")
        r.u("I")
        itoks = {}
        for _ in range(m):
            if not pair_seen:
                r.ci()
                pair_seen = True
            idx = r.u("Q")
            itoks[idx] = r.s()
        real.append((ts, itoks))
    out["real"] = real
    out["itok_bucket_fields"] = bucket_fields
    out["max_tokens"] = r.u("Q") if sai_version >= 1 else 300
    out["consumed"], out["size"] = r.p, len(r.d)
    return out
