import sys, ctypes as C, threading, time
sys.path.insert(0,'rs-face-detection_amd/python'); sys.path.insert(0,'tests')
import numpy as np, helpers, rfd_hip
H=C.CDLL('tools/bin/liblds_dma_hazard.so')
n=8
bb=rfd_hip.BACKBONE_MNET025 if sys.argv[1]=='mnet' else rfd_hip.BACKBONE_R50
op=int(sys.argv[2])
dist = rfd_hip.RetinaFaceDetection(max_batch_size=n, max_det=2048, confidence_threshold=0.3, backbone=bb)
dist.init_synthetic_weights(1234)
dist.debug_set_concurrency(False, 8, 1, False)
dist.call_batch([helpers.make_image(i,640,640,n_blobs=3) for i in range(n)])
H.hazard_f3_init()
names={0:'broadcast b128',7:'two-address b128',1:'per-lane b128',5:'fragment b128',4:'broadcast b96',2:'broadcast b64',6:'per-lane b64',3:'broadcast b32',8:'8 reads in flight',9:'8 reads + vmem'}
out=(C.c_ulonglong*64)()
for disturb in (False, True):
    stop=[False]
    def loop():
        while not stop[0]: dist.debug_run(n,op,op)
    if disturb:
        th=threading.Thread(target=loop); th.start(); time.sleep(0.05)
    print('disturber: conv op %d of %s'%(op,sys.argv[1]) if disturb else 'disturber: none')
    for shp,nm in names.items():
        H.hazard_victim(shp, 8, 2000, out)
        q=[sum(out[i*16:(i+1)*16]) for i in range(4)]
        print('  %-20s bad reads %10d by lane quarter %s'%(nm,sum(q),q),flush=True)
    for var,nm in ((0,'first-conv kernel as shipped then'),(1,'same, LDS allocation padded to 8 KiB'),(2,'same, table 4 KiB into the allocation'),(3,'explicit b128 reads, all landed before the math'),(4,'explicit 2 x b64 reads, all landed'),(5,'explicit b128 reads, math under counted waits')):
        H.hazard_f3_run(var, 12, out)
        q=[sum(out[i*16:(i+1)*16]) for i in range(4)]
        print('  %-44s bad elements %8d by lane quarter %s'%(nm,sum(q),q),flush=True)
    if disturb:
        stop[0]=True; th.join()
