import torch, time
dev="cuda:0"
B,R,Rcap,N1,E=64,12100,20200,101,128
DL=torch.randn(B,R,N1,device=dev); Ofull=torch.randn(B,Rcap,E,device=dev); O=Ofull[:,:R]; dQ=torch.randn(B,R,E,device=dev)
def t(f,n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time()-t0)/n*1e3
print("bmm(DL^T, O strided)      ", t(lambda: torch.bmm(DL.transpose(1,2), O)))
print("bmm(DL^T, dQ contiguous)  ", t(lambda: torch.bmm(DL.transpose(1,2), dQ)))
print("bmm(O^T strided, DL)^T    ", t(lambda: torch.bmm(O.transpose(1,2), DL)))
print("bmm(dQ^T, DL)             ", t(lambda: torch.bmm(dQ.transpose(1,2), DL)))
print("O.contiguous()            ", t(lambda: O.contiguous()))
print("einsum brn,bre->bne       ", t(lambda: torch.einsum("brn,bre->bne", DL, O)))
PK=torch.randn(B,N1,E,device=dev)
print("dO = bmm(DL, PK)          ", t(lambda: torch.bmm(DL, PK)))
print("DL.sum(1)                 ", t(lambda: DL.sum(dim=1)))
cat=torch.cat([O, torch.ones(B,R,1,device=dev)],2)
