"""Hash of the machine code of every kernel of a HIP source (device-only compile to assembly; comments and directives dropped):
shows that a source clean-up left the product kernels' instructions unchanged.
    python tools/isa_hash.py font_ocr_amd/csrc/hip/scan_mfma2.hip [out.json]"""
import hashlib, json, os, re, subprocess, sys, tempfile
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = sys.argv[1]
with tempfile.TemporaryDirectory() as d:
    asm = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", f"-I{root}/include", f"-I{root}/font_ocr_amd/csrc/hip",
                    "--cuda-device-only", "-S", "-o", asm, src], check=True, stderr=subprocess.DEVNULL)
    text = open(asm).read()
out = {}
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s+s_endpgm", text, re.S | re.M):
    body = "\n".join(l.split(";")[0].rstrip() for l in m.group(2).split("\n") if l.strip() and not l.strip().startswith((";", ".")))
    out[m.group(1)] = hashlib.sha256(body.encode()).hexdigest()[:16]
for n, h in sorted(out.items()):
    print(h, n[:110])
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=0, sort_keys=True)
