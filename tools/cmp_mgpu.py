"""Compare two outputs of tools/mgpu_check.py (rows of accel[3], GravPM[3], Potential): usage cmp_mgpu.py a.npy b.npy label"""
import numpy as np, sys
a=np.load(sys.argv[1]); b=np.load(sys.argv[2])
print(sys.argv[3], "acc maxdiff %.3e  gravpm: mean|one| %.3e maxdiff %.3e  pot maxdiff %.3e" % (np.abs(a[:,0:3]-b[:,0:3]).max(), np.abs(a[:,3:6]).mean(), np.abs(a[:,3:6]-b[:,3:6]).max(), np.abs(a[:,6]-b[:,6]).max()))
