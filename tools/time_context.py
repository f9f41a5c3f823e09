import importlib, time, sys
sys.path.insert(0, "/root/repo")
t=time.time(); kzg = importlib.import_module("rust-eth-kzg_amd"); kzg.load_library(); print("load lib", time.time()-t)
t=time.time(); c = kzg.DASContext(True); print("ctx", time.time()-t)
t=time.time(); c.close(); print("close", time.time()-t)
t=time.time(); c = kzg.DASContext(True); print("ctx2", time.time()-t)
