"""Share of binomial rejection attempts the table-free first phase settles, per (n, p) (developer tool; testing build)."""
import ctypes as C, numpy as np, sys
sys.path.insert(0,'.')
import epidemicmodeling_amd as em
from epidemicmodeling_amd import _lib
_lib._lib=_lib.load_testing()
ctx = em.interface.context_for(em.make_pomdp(LOD=5))
out=(C.c_uint64*5)()
for lo,span,pp in [(20,40,0.5),(2000,1,0.0051),(2000,1,0.035),(2000,1,0.127),(2000,1,0.245),(2000,1,0.44),(10000,1,0.0025),(10000,1,0.035),(10000,1,0.1),(100000,900000,0.5),(1<<30,1<<30,0.37)]:
    rc=ctx.lib.epi_debug_btrs_quick(ctx.handle,5,lo,span,pp,1650,out)
    t=out[0]
    print(lo,span,pp,"sigma %.1f"%np.sqrt((lo+span/2)*pp*(1-pp)),"acc %.4f rej %.4f open %.4f bad %d"%(out[1]/t,out[2]/t,out[3]/t,out[4]))
