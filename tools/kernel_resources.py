"""Developer tool: per-kernel register / LDS / scratch / code-size table of the gfx950 device code (no GPU needed).
   python tools/kernel_resources.py [filter] [-- extra hipcc flags]"""
import importlib.util
import os
import re
import subprocess
import sys
import tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("agpt_build", os.path.join(root, "ag-pathtracer_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
args = sys.argv[1:]
extra = []
if "--" in args:
    k = args.index("--")
    args, extra = args[:k], args[k + 1:]
flt = args[0] if args else ""
flags = [f for f in b.FLAGS if f not in ("-shared", "-fPIC")]
text = ""
with tempfile.TemporaryDirectory() as tmp:
    for src in b.SOURCES:
        if not src.endswith(".hip"):
            continue
        asm = os.path.join(tmp, src + ".s")
        cmd = ["/opt/rocm/bin/hipcc"] + flags + b.SOURCE_FLAGS.get(src, []) + extra + ["--cuda-device-only", "-S", "-o", asm, os.path.join(b.CSRC, src)]
        subprocess.check_call(cmd)
        text += open(asm).read() + "\n"
if os.environ.get("KEEP_ASM"):
    open(os.environ["KEEP_ASM"], "w").write(text)
demangle = subprocess.run(["c++filt"], input="\n".join(re.findall(r"^\s*\.amdhsa_kernel (\S+)", text, re.M)),
                          capture_output=True, text=True).stdout.split("\n")
names = re.findall(r"^\s*\.amdhsa_kernel (\S+)", text, re.M)
pretty = dict(zip(names, demangle))
print("%-72s %5s %5s %6s %7s %7s %6s" % ("kernel", "vgpr", "sgpr", "spill", "scratch", "lds", "insts"))
for n in names:
    body = text.split(".amdhsa_kernel " + n, 1)[1].split(".end_amdhsa_kernel", 1)[0]
    g = lambda key: (re.search(r"\.amdhsa_%s (\S+)" % key, body) or [None, "?"])[1]
    # the code itself: from the label to its .Lfunc_end
    m = re.search(r"^%s:[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(n), text, re.M | re.S)
    code = m.group(1) if m else ""
    insts = len([l for l in code.split("\n") if re.match(r"^\s+[a-z_0-9]+", l) and not l.strip().startswith(".")])
    meta = re.search(r"\.name:\s+%s\n(.*?)\.wavefront_size" % re.escape(n), text, re.S)
    mt = meta.group(1) if meta else ""
    mg = lambda key: (re.search(r"\.%s:\s+(\S+)" % key, mt) or [None, "?"])[1]
    p = re.sub(r"\(.*", "", pretty.get(n, n)).replace("void ", "")
    if flt and flt not in p:
        continue
    print("%-72s %5s %5s %6s %7s %7s %6d" % (p[:72], mg("vgpr_count"), mg("sgpr_count"), mg("vgpr_spill_count"), mg("private_segment_fixed_size"),
                                           g("group_segment_fixed_size"), insts))
