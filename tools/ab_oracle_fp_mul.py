import time, sys, os, subprocess
root=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0, os.path.join(root,"tests"))
import oracle_lib, synth
src=[os.path.join(root,"oracle",f) for f in ("field.c","g1.c","pairing.c","sha256.c","kzg.c")]
for tag,flags in (("portable",[]),("adx",["-DORACLE_ADX"]),("portable",[]),("adx",["-DORACLE_ADX"])):
    so="/tmp/liboracle_%s.so"%tag
    subprocess.check_call(["gcc","-O3","-march=native",*flags,"-fopenmp","-fPIC","-std=gnu11","-shared","-o",so]+src)
    oracle_lib._SO=so
    o=oracle_lib.Oracle(use_precomp=True, threads=1)
    b=synth.seeded_blob(5)
    o.compute_cells_and_kzg_proofs(b)
    t=time.time()
    for i in range(4): o.compute_cells_and_kzg_proofs(b)
    print(tag, round((time.time()-t)/4*1e3,1),"ms per blob single-thread", flush=True)
    o.close()
