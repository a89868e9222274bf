"""Developer tool: build an A/B variant of the library, `libagpt_hip_<tag>.so`, with extra hipcc flags.
   python tools/build_variant.py <tag> [extra flags...] [--shade flags for agpt_shade_kernels.hip only...];
   select at run time with AGPT_LIB_VARIANT=<tag>."""
import importlib.util
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("agpt_build", os.path.join(root, "ag-pathtracer_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
tag, extra = sys.argv[1], sys.argv[2:]
shade = []
if "--shade" in extra:
    k = extra.index("--shade")
    extra, shade = extra[:k], extra[k + 1:]
out = os.path.join(b.HERE, "libagpt_hip_%s.so" % tag)
b.build_library(out, extra=extra, extra_for={"agpt_shade_kernels.hip": shade})
print(out)
