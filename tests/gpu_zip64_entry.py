"""One-off (GPU box): a Zip_64 archive with a 4.2 GiB entry (Zip.Create promotes the format: zip-create.adb:161-179, 237-251,
682-752) written by ZipCreate with the GPU encoder (zada_compress_data on host buffers, span after span), read back by
Python's zipfile."""
import importlib, io, os, sys, time, zipfile, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
za = importlib.import_module("zip-ada_amd")
n = (int(sys.argv[1]) << 20) if len(sys.argv) > 1 else (4 << 30) + (200 << 20)
big = za.silesia_mix(n)
enc = za.Encoder(0)
zc = za.ZipCreate(enc, za.Method.Deflate_1)
t0 = time.time()
zc.add_stream("small/first.txt", za.silesia_mix(100000, class_mask=1).tobytes())
zc.add_stream("big.bin", big)
zc.add_stream("small/last.txt", b"the end\n")
arc = zc.finish()
print("archive of %d bytes written in %.1f s, Zip_64: %s" % (len(arc), time.time() - t0, zc.zip64))
zf = zipfile.ZipFile(io.BytesIO(arc))
infos = zf.infolist()
print([(i.filename, i.file_size, i.compress_size) for i in infos])
crc = 0
with zf.open("big.bin") as f:
    while True:
        ch = f.read(1 << 24)
        if not ch:
            break
        crc = zlib.crc32(ch, crc)
print("big entry reads back with the right CRC:", crc == zlib.crc32(big) == infos[1].CRC, "; last entry:", zf.read("small/last.txt"))
