"""Minimal BGZF/BAM writer for hand-made known-answer inputs (tests only)."""
import struct
import zlib

_OPS = "MIDNSHP=X"
_BASES = "=ACMGRSVTWYHKDBN"


def _bgzf_block(data: bytes) -> bytes:
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = 18 + len(comp) + 8 - 1
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def parse_cigar(s):
    out, num = [], ""
    for ch in s:
        if ch.isdigit():
            num += ch
        else:
            out.append((ch, int(num)))
            num = ""
    return out


def record(name, refid, pos, mapq, flag, cigar, mrefid=-1, mpos=-1, seq=None, qual=None, tags=b"NHC\x01", lseq=None):
    """`lseq`: store fewer bases than the CIGAR consumes (a secondary line with SEQ cut short or '*': the reference's assert, ReadRec.cpp:64)"""
    cig = parse_cigar(cigar) if isinstance(cigar, str) else cigar
    if lseq is None:
        lseq = sum(l for op, l in cig if op in "MIS=X")
    if seq is None:
        seq = "ACGT" * (lseq // 4 + 1)
        seq = seq[:lseq]
    if qual is None:
        qual = [30] * lseq
    body = struct.pack("<iiBBHHHiiii", refid, pos, len(name) + 1, mapq, 4680, len(cig), flag, lseq, mrefid, mpos, 0)
    body += name.encode() + b"\0"
    for op, l in cig:
        body += struct.pack("<I", (l << 4) | _OPS.index(op))
    packed = bytearray((lseq + 1) // 2)
    for i, b in enumerate(seq):
        packed[i >> 1] |= _BASES.index(b) << (4 if i % 2 == 0 else 0)
    body += bytes(packed) + bytes(qual) + tags
    return struct.pack("<i", len(body)) + body


def write_bam(path, contigs, records, sort_order="coordinate"):
    text = f"@HD\tVN:1.4\tSO:{sort_order}\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in contigs)
    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(contigs))
    for n, l in contigs:
        hdr += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    data = hdr + b"".join(records)
    with open(path, "wb") as f:
        for i in range(0, len(data), 0xff00):
            f.write(_bgzf_block(data[i:i + 0xff00]))
        f.write(_EOF)
