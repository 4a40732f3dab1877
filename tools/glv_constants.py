"""BN254 constants of the GLV endomorphism used by csrc/gbasis.hip, derived and checked numerically here: cube roots of unity beta (mod q) and
lambda (mod r) with phi(x, y) = (beta x, y) = lambda (x, y) on G1 and on the order-r subgroup of the twist, and the short lattice basis."""
q=21888242871839275222246405745257275088696311157297823662689037894645226208583
r=21888242871839275222246405745257275088548364400416034343698204186575808495617
def cube_roots(p):
    # nontrivial cube roots of unity mod p
    g=2
    while True:
        x=pow(g,(p-1)//3,p)
        if x!=1: return x, x*x%p
        g+=1
b1_,b2_=cube_roots(q); l1_,l2_=cube_roots(r)
# G1 affine arithmetic
def inv(a,p): return pow(a,p-2,p)
def add1(P,Q):
    if P is None: return Q
    if Q is None: return P
    x1,y1=P;x2,y2=Q
    if x1==x2:
        if (y1+y2)%q==0: return None
        l=3*x1*x1*inv(2*y1,q)%q
    else: l=(y2-y1)*inv(x2-x1,q)%q
    x3=(l*l-x1-x2)%q; return (x3,(l*(x1-x3)-y1)%q)
def mul1(k,P):
    R=None
    for b in bin(k)[2:]:
        R=add1(R,R)
        if b=='1': R=add1(R,P)
    return R
G=(1,2)
for lam in (l1_,l2_):
    P=mul1(lam,G)
    for beta in (b1_,b2_):
        if P==(beta*G[0]%q,G[1]): print("G1: lambda",hex(lam),"beta",hex(beta)); LAM=lam; BETA=beta
# Fq2 arithmetic
class F2:
    def __init__(s,a,b=0): s.a=a%q; s.b=b%q
    def __add__(s,o): return F2(s.a+o.a,s.b+o.b)
    def __sub__(s,o): return F2(s.a-o.a,s.b-o.b)
    def __mul__(s,o):
        if isinstance(o,int): return F2(s.a*o,s.b*o)
        return F2(s.a*o.a-s.b*o.b,s.a*o.b+s.b*o.a)
    def inv(s):
        n=inv((s.a*s.a+s.b*s.b)%q,q); return F2(s.a*n,-s.b*n)
    def __eq__(s,o): return s.a==o.a and s.b==o.b
    def iszero(s): return s.a==0 and s.b==0
def add2(P,Q):
    if P is None: return Q
    if Q is None: return P
    x1,y1=P;x2,y2=Q
    if x1==x2:
        if (y1+y2).iszero(): return None
        l=(x1*x1*3)*((y1*2).inv())
    else: l=(y2-y1)*((x2-x1).inv())
    x3=l*l-x1-x2; return (x3,l*(x1-x3)-y1)
def mul2(k,P):
    R=None
    for b in bin(k)[2:]:
        R=add2(R,R)
        if b=='1': R=add2(R,P)
    return R
G2=(F2(10857046999023057135944570762232829481370756359578518086990519993285655852781,11559732032986387107991004021392285783925812861821192530917403151452391805634),
    F2(8495653923123431417604973247489272438418190587263600148770280649306958101930,4082367875863433681332203403145435568316851327593401208105741076214120093531))
assert mul2(r,G2) is None
Q=mul2(LAM,G2)
for beta in (b1_,b2_):
    if Q[0]==G2[0]*beta and Q[1]==G2[1]: print("G2: lambda (same as G1) pairs with beta",hex(beta)); BETA2=beta
# lattice basis by extended Euclid on (r, LAM)
import math
def glv_basis(n,lam):
    s0,t0,r0=1,0,n; s1,t1,r1=0,1,lam
    rs=[(r0,t0),(r1,t1)]
    while r1!=0:
        qq=r0//r1
        r0,r1=r1,r0-qq*r1; t0,t1=t1,t0-qq*t1
        rs.append((r1,t1))
    sq=math.isqrt(n)
    # find l: r_l >= sqrt(n) > r_{l+1}
    for i in range(len(rs)-1):
        if rs[i][0]>=sq and rs[i+1][0]<sq: l=i;break
    a1,b1=rs[l+1][0],-rs[l+1][1]
    c0=(rs[l][0],-rs[l][1]); c2=(rs[l+2][0],-rs[l+2][1])
    a2,b2 = c0 if c0[0]**2+c0[1]**2 <= c2[0]**2+c2[1]**2 else c2
    return a1,b1,a2,b2
a1,b1,a2,b2=glv_basis(r,LAM)
assert (a1+b1*LAM)%r==0 and (a2+b2*LAM)%r==0
print("a1",a1,"b1",b1,"a2",a2,"b2",b2, [x.bit_length() for x in (a1,b1,a2,b2)])
det=a1*b2-a2*b1; print("det",det==r, det==-r)
SH=256
g1=(b2<<SH)//det if det>0 else ((-b2)<<SH)//(-det)
import random
mx=0
def decomp(k):
    # c1 = round(b2 k / det), c2 = round(-b1 k / det)
    c1=(b2*k*2+det)//(2*det) if det>0 else None
    c2=(-b1*k*2+det)//(2*det)
    k1=k-c1*a1-c2*a2; k2=-c1*b1-c2*b2
    return k1,k2
for _ in range(2000):
    k=random.randrange(r); k1,k2=decomp(k)
    assert (k1+k2*LAM-k)%r==0
    mx=max(mx,abs(k1).bit_length(),abs(k2).bit_length())
print("max bits exact rounding",mx)
print("LAM",hex(LAM)); print("BETA",hex(BETA)); print("BETA2",hex(BETA2))
