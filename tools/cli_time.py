import subprocess, time, sys, os
R="/tmp/yaha_bench_cache/g100m_n16384_l1000_s1000.fa"; X="/tmp/yaha_bench_cache/g100m_s42.X15_01_65525S"
outs=[]
for t in (8, 32, 64):
    o="/tmp/out_%d.sam"%t; s=time.time()
    subprocess.run([os.path.join(os.environ["GRAFT_REPO_ROOT"],"yaha_amd/csrc/yaha"),"-x",X,"-q",R,"-osh",o,"-t",str(t),"-batch","8192"],stderr=subprocess.DEVNULL,check=True)
    dt=time.time()-s; print("t=%d wall %.2f s -> %.0f reads/s end to end (incl. index mmap+upload)"%(t,dt,16384/dt)); outs.append(o)
a=[l for l in open(outs[0]) if not l.startswith("@PG")]; b=[l for l in open(outs[2]) if not l.startswith("@PG")]
print("identical minus @PG:", a==b, len(a))
