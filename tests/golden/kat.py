"""Known-answer vectors transcribed from the reference's own tests/doctests (DATA only).

Every entry cites the reference file:line (relative to sile/libflate v2.3.0) that holds it.
They pin the oracle (tests/test_oracle_kat.py) and, through the same tables, the HIP path
(tests/test_gpu_parity.py::test_kat_*).
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def _read(*parts):
    with open(os.path.join(HERE, *parts), "rb") as f:
        return f.read()


HELLO = b"Hello World!"

# ---- encoder pins (bit-exact) -------------------------------------------------------------
# src/deflate/encode.rs:152-154 and :199-201 — deflate::Encoder::new, one write_all
DEFLATE_HELLO = bytes([5, 192, 49, 13, 0, 0, 8, 3, 65, 43, 224, 6, 7, 24, 128, 237, 147, 38, 245,
                       63, 244, 230, 65, 181, 50, 215, 1])
# src/deflate/encode.rs:178-180 — EncodeOptions::new().no_compression()
DEFLATE_HELLO_STORED = bytes([1, 12, 0, 243, 255, 72, 101, 108, 108, 111, 32, 87, 111, 114, 108,
                              100, 33])
# src/zlib.rs:547-549 and :610-612 — zlib::Encoder::new
ZLIB_HELLO = bytes([120, 156, 5, 192, 49, 13, 0, 0, 8, 3, 65, 43, 224, 6, 7, 24, 128, 237, 147, 38,
                    245, 63, 244, 230, 65, 181, 50, 215, 1, 28, 73, 4, 62])
# src/zlib.rs:573-575 and :750-753 — zlib no_compression
ZLIB_HELLO_STORED = bytes([120, 1, 1, 12, 0, 243, 255, 72, 101, 108, 108, 111, 32, 87, 111, 114,
                           108, 100, 33, 28, 73, 4, 62])
# src/gzip.rs:800-802 — gzip no_compression, mtime 123
GZIP_HELLO_STORED = bytes([31, 139, 8, 0, 123, 0, 0, 0, 0, 3, 1, 12, 0, 243, 255, 72, 101, 108,
                           108, 111, 32, 87, 111, 114, 108, 100, 33, 163, 28, 41, 28, 12, 0, 0, 0])
# src/zlib.rs:840-902 test_issues_27 — writes + flush() twice + finish
ISSUE27_WRITES = [b"fooooooooooooooooo", b"bar", b"baz"]
ISSUE27_PLAIN = b"fooooooooooooooooobarbazfooooooooooooooooobarbaz"
ISSUE27_ZLIB_NONE = bytes([
    120, 156,
    92, 192, 161, 17, 0, 0, 0, 1, 192, 89, 9, 170, 59, 209, 244, 186, 151, 31, 17, 162,
    227, 2, 14, 141, 0, 0, 0, 8, 0, 206, 74, 80, 221, 137, 166, 215, 189, 252, 136, 16, 93,
    1, 112, 32, 0, 0, 0, 0, 0, 228, 255, 26, 246, 95, 20, 111])
ISSUE27_ZLIB_SYNC = bytes([
    120, 156,
    92, 192, 161, 17, 0, 0, 0, 1, 192, 89, 9, 170, 59, 209, 244, 186, 151, 31, 17, 162, 3,
    0, 0, 255, 255,
    92, 192, 161, 17, 0, 0, 0, 1, 192, 89, 9, 170, 59, 209, 244, 186, 151, 31, 17, 162, 3,
    0, 0, 255, 255,
    5, 192, 129, 0, 0, 0, 0, 0, 144, 255, 107, 0, 246, 95, 20, 111])
# src/lz77.rs:16-32 — DefaultLz77Encoder on b"aaaaa": [Literal 'a', Pointer{4,1}]
LZ77_AAAAA = [(97, 0), (4, 1)]
# src/bit.rs:182-194 — BitWriter: 1, 3 bits 0b010, 11 bits 0b10101011010, flush, 1, flush
BITWRITER_OUT = bytes([0b10100101, 0b01010101, 0b00000001])
# src/checksum.rs:44-56
CRC32_ABCDE = 0x8587D865
ADLER32_ABCDE = 0x05C801F0

# ---- decoder pins -------------------------------------------------------------------------
# src/deflate/decode.rs:28,60 — fixed-Huffman "Hello World!"
DEFLATE_HELLO_FIXED = bytes([243, 72, 205, 201, 201, 87, 8, 207, 47, 202, 73, 81, 4, 0])
# src/zlib.rs:708-728
ZLIB_HELLO_FIXED = bytes([120, 156, 243, 72, 205, 201, 201, 87, 8, 207, 47, 202, 73, 81, 4, 0, 28,
                          73, 4, 62])
# src/gzip.rs:1072-1083 — two members made by an OLDER encoder (empty distance table)
GZIP_MEMBER_HELLO_ = bytes([31, 139, 8, 0, 51, 206, 75, 90, 0, 3, 5, 128, 49, 9, 0, 0, 0, 194, 170,
                            24, 199, 34, 126, 3, 251, 127, 163, 131, 71, 192, 252, 45, 234, 6, 0,
                            0, 0])
GZIP_MEMBER_WORLD = bytes([31, 139, 8, 0, 227, 207, 75, 90, 0, 3, 5, 128, 49, 9, 0, 0, 0, 194, 178,
                           152, 202, 2, 158, 130, 96, 255, 99, 120, 111, 4, 222, 157, 40, 118, 6,
                           0, 0, 0])
# data/noncompressed_block_offset_sync (src/non_blocking/gzip.rs:177-183)
OFFSET_GZ = _read("noncompressed_block_offset_sync", "offset.gz")
OFFSET_PLAIN = _read("noncompressed_block_offset_sync", "offset")

# ---- reject pins --------------------------------------------------------------------------
# src/deflate/decode.rs:175-192 test_issues_3: dynamic table must LOAD fine (first 3 bits + table)
ISSUE3_INPUT = bytes([
    180, 253, 73, 143, 28, 201, 150, 46, 8, 254, 150, 184, 139, 75, 18, 69, 247, 32, 157,
    51, 27, 141, 132, 207, 78, 210, 167, 116, 243, 160, 223, 136, 141, 66, 205, 76, 221,
    76, 195, 213, 84, 236, 234, 224, 78, 227, 34, 145, 221, 139, 126, 232, 69, 173, 170,
    208, 192, 219, 245, 67, 3, 15, 149, 120, 171, 70, 53, 106, 213, 175, 23, 21, 153, 139,
    254, 27, 249, 75, 234, 124, 71, 116, 56, 71, 68, 212, 204, 121, 115, 64, 222, 160, 203,
    119, 142, 170, 169, 138, 202, 112, 228, 140, 38])
# src/deflate/decode.rs:194-212 it_works: InvalidData, message starts "Too long backword reference"
TOO_LONG_BACKREF = ISSUE3_INPUT + bytes([171, 162, 88, 212, 235, 56, 136, 231, 233, 239, 113, 249,
                                         163, 252, 16, 42, 138, 49, 226, 108, 73, 28, 153])
# src/deflate/decode.rs:214-220 test_issue_64: must be an error (not a panic)
ISSUE64 = b"\x04\x04\x04\x05:\x1az*\xfc\x06\x01\x90\x01\x06\x01"
# src/gzip.rs:1228-1247 issue_15_{1,2,3}: must be errors
ISSUE15_1 = (b"\x1F\x8B\x08\xC1\x7B\x7B\x7B\x7B\x7B\x7B\x7B\x7B\x7B\x7B\x7B\x7B\x7B\x7B\x7B\x7B\x7B"
             b"\x7B\x80\x80\x80\x80\x7B\x7B\x7B\x7B\x7B\x7B\x97\x7B\x7B\x7B\x86\x27\xEB\x60\xA7\xA8"
             b"\x46\x6E\x1F\x33\x51\x5C\x34\xE0\xD2\x2E\xE8\x0C\x19\x1D\x3D\x3C\xFD\x3B\x6A\xFA\x63"
             b"\xDF\x28\x87\x86\xF2\xA6\xAC\x87\x86\xF2\xA6\xAC\xD5")
ISSUE15_2 = (b"\x1F\x8B\x08\xC1\x7B\x7B\x7B\x7B\x7B\xFC\x5D\x2D\xDC\x08\xC1\x7B\x7B\x7B\x7B\x7B\xFC"
             b"\x5D\x2D\xDC\x08\xC1\x7B\x7F\x7B\x7B\x7B\xFC\x5D\x2D\xDC\x69\x32\x48\x22\x5A\x81\x81"
             b"\x42\x42\x81\x7E\x81\x81\x81\x81\xF2\x17")
ISSUE15_3 = (b"\x1F\x8B\x08\xC1\x91\x28\x71\xDC\xF2\x2D\x34\x35\x31\x35\x34\x30\x70\x6E\x60\x35\x31"
             b"\x32\x32\x33\x32\x33\x37\x32\x36\x38\xDD\x1C\xE5\x2A\xDD\xDD\xDD\x22\xDD\xDD\xDD\xDC"
             b"\x88\x13\xC9\x40\x60\xA7")
# data/issues_16/* (src/zlib.rs:798-837): message prefix "The value of HDIST is too big: max=30"
ISSUES_16 = [_read("issues_16", f) for f in sorted(os.listdir(os.path.join(HERE, "issues_16")))]
# src/zlib.rs:916-934 issue71: truncated → error, 33 bytes recoverable
ISSUE71_IN = bytes([120, 218, 251, 255, 207, 144, 193, 138, 193, 151, 161, 146, 33, 143, 33, 149,
                    161, 156, 161, 24, 72, 38, 51, 148, 48, 100, 50, 228, 3, 69, 120, 25, 184, 24])
ISSUE71_OUT = bytes([255, 254, 49, 0, 58, 0, 77, 0, 121, 0, 110, 0, 101, 0, 119, 0, 115, 0, 101, 0,
                     99, 0, 116, 0, 105, 0, 111, 0, 110, 0, 13, 0, 10])
# src/zlib.rs:936-943 issue_82: header [0,0] → InvalidData, message contains "method=0"
ISSUE82 = bytes([0, 0])
# src/zlib.rs:700-706 test_issue_2 — round-trip inputs
ISSUE2_INPUTS = [
    bytes([163, 181, 167, 40, 62, 239, 41, 125, 189, 217, 61, 122, 20, 136, 160, 178, 119, 217,
           217, 41, 125, 189, 97, 195, 101, 47, 170]),
    bytes([162, 58, 99, 211, 7, 64, 96, 36, 57, 155, 53, 166, 76, 14, 238, 66, 66, 148, 154, 124,
           162, 58, 99, 188, 138, 131, 171, 189, 54, 229, 192, 38, 29, 240, 122, 28]),
    bytes([239, 238, 212, 42, 5, 46, 186, 67, 122, 247, 30, 61, 219, 62, 228, 202, 164, 205, 139,
           109, 99, 181, 99, 181, 99, 122, 30, 12, 62, 46, 27, 145, 241, 183, 137]),
    bytes([88, 202, 64, 12, 125, 108, 153, 49, 164, 250, 71, 19, 4, 108, 111, 108, 237, 205, 208,
           77, 217, 100, 118, 49, 10, 64, 12, 125, 51, 202, 69, 67, 181, 146, 86]),
]
# src/deflate/test_data.rs ISSUE_52_INPUT (encode.rs:434-457: compressed must be smaller)
ISSUE52 = _read("issue_52_input.bin")

# ---- secondary pins (scratch restatement; NOT reference-produced) ------
# (name, write_size (0 = S1), N, C, sha256 of the raw-DEFLATE output)
def ramp():
    return bytes(i & 0xFF for i in range(32768 * 32))  # src/deflate/mod.rs:50-52


def test_i():
    return b"".join(b"test %d" % i for i in range(10000))  # non_blocking/deflate/decode.rs:274-277


SECONDARY = [
    ("empty", lambda: b"", 0, 12, "699532b3a0e4e9218297e6636e1dd47449a5347a564fd5ff468871c14d3714ff"),
    ("a", lambda: b"a", 0, 13, "5a2584da7e3f0161dceba9577f11e2e53d067b36ffa4f80298dd42139a238a6f"),
    ("aaaaa", lambda: b"aaaaa", 0, 15, "46fd555484c360cd6e52b764da83899365f400cbf847e3ac83d1b32431900b86"),
    ("hello3", lambda: b"hello hello hello", 0, 22, "f37843df26ef93dfb1906c29b2eb6ea64612fff3af93d371082604a639f641f7"),
    ("issue52_16031", lambda: ISSUE52[:16031], 0, 2707, "c605da79e8ba5d0a2733af04c173a8ae0d305eb1630bec97226550ea64597016"),
    ("issue52_16032", lambda: ISSUE52[:16032], 0, 2708, "065987f10853e88cff9136795e34c156776ac24130482cf5ad9ef2f3c5b42007"),
    ("issue52_full", lambda: ISSUE52, 0, 2716, "3dccd92b74e42c45592cf877096cab07b1a3c023a3c59f19f2b1b7b7605647e9"),
    ("ramp_s1", ramp, 0, 4397, "33baae8ab95272fb21ac5be30ce1fb5d5cbfea5b52a2909382ef7757368c625e"),
    ("ramp_s8k", ramp, 8192, 5266, "9148724c14e1437ece9a356325e2fb71333b1352213beced513903156f93321f"),
    ("zeros300k", lambda: bytes(300000), 0, 307, "a32dd9216e68e8151d463bd21eabf4e54f16b7a80828b6fd6465dc3e463a62ba"),
    ("test_i", test_i, 0, 18702, "8274f6b591735981ebaeab16954def56ce648fcbaeccedbe80ee19dbf8f74e09"),
]
