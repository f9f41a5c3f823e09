"""Context creation timing (ETH_KZG_AMD_TRACE=1 prints the stages): first context, close, second context."""
import importlib, os, time, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t=time.time(); kzg = importlib.import_module("rust-eth-kzg_amd"); kzg.load_library(); print("load lib", time.time()-t)
t=time.time(); c = kzg.DASContext(True); print("ctx", time.time()-t)
t=time.time(); c.close(); print("close", time.time()-t)
t=time.time(); c = kzg.DASContext(True); print("ctx2", time.time()-t)
