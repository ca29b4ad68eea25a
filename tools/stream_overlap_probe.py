import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
N=8192
P=bench.build_pipeline(N,0.5,"f32",torch)
q,eng=P["q"],P["eng"]
nids=P["nids"]; norm=P["geom"].area/float(N*N)**2
tmaps=[eng.irfft(eng.grf_hc(1234,i,P["cs"]),scale=1.0/np.sqrt(eng.npix)) for i in range(2)]
def run(nstreams, steps=24):
    qs=[q]+[q.fork() for _ in range(nstreams-1)]
    streams=[torch.cuda.Stream() for _ in range(nstreams)]
    kT=[e.eng.hc() for e in qs]; kk=[e.eng.hc() for e in qs]
    def step(i):
        j=i%nstreams
        with torch.cuda.stream(streams[j]):
            e=qs[j].eng
            e.rfft(tmaps[i&1],out=kT[j]); qs[j].reconstruct_tt_hc(kT[j],out=kk[j])
            s,c=e.bin_power(kk[j],kk[j],norm,P["ids"],nids,herm=True)
        return s
    for i in range(4): step(i)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(steps): last=step(i)
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    return steps/dt, last
for ns in (1,2,3,1,2):
    r,last=run(ns); print(ns,"streams:", "%.1f recon/s"%r, last[3:6].cpu().numpy())
