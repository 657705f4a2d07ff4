"""Builds a variant of libpdx_hip.so with extra -D flags for groupby.hip into tools/_ab/ (git-ignored; travels to the GPU box):
   python tools/build_variant.py tm -DPDX_FLR_TIMING=1     ->  tools/_ab/libpdx_tm.so, loaded with PDX_LIB_PATH=...
A/B runs of one box alternate between such a library and the product build."""
import sys, os, subprocess
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
name, flags = sys.argv[1], sys.argv[2:]
objs=[os.path.join(g.CSRC,s.replace('.hip','.o')) for s in g.HIP_SOURCES if s!='groupby.hip']
os.makedirs('/root/repo/tools/_ab', exist_ok=True)
o=f'/root/repo/tools/_ab/groupby_{name}.o'
subprocess.check_call(['/opt/rocm/bin/hipcc',*g.HIPCC_FLAGS,*flags,'-c',os.path.join(g.CSRC,'groupby.hip'),'-o',o],stderr=subprocess.DEVNULL)
subprocess.check_call(['/opt/rocm/bin/hipcc','--offload-arch=gfx950','-shared','-fPIC','-o',f'/root/repo/tools/_ab/libpdx_{name}.so',*objs,o,'-ldl'])
print('built', name)
