"""Offline accounting of the causal attention kernel's tile schedule per compact class."""
import numpy as np, sys
def sched(Actx, QB=128, WQ=32, KT=64, T=32, plain=False):
    if plain:
        Ar=Actx; A3=3*Ar; Lreg=T*A3; rep_keys=0; Lq=Lreg
    else:
        Ar=Actx-1; A3=3*Ar; Lreg=T*A3; rep_keys=3*T; Lq=Lreg+rep_keys
    Lk=Lreg; rep_pos0=Lreg
    nqb=(Lq+QB-1)//QB; nw=QB//WQ
    tot=dict(unm=0,msk=0,tiles_wg=0,wave_tiles=0,wgs=nqb,idle_wave_tiles=0, crit=0.0)
    useful=0
    # useful pairs
    for pos in range(Lq):
        if pos>=rep_pos0:
            tq=(pos-rep_pos0)//3; kq=(pos-rep_pos0)%3
            useful+= tq*A3 + Ar + (3*tq+1) + kq   # regular earlier + state tokens of step + rep keys visible (m-fold counted once) + own
        else:
            tq=pos//A3; rem=pos%A3; aq=rem//3; kq=rem%3
            useful+= tq*A3 + Ar + kq + (0 if kq==0 else 0) + ((3*tq+1) if rep_keys else 0)
            # own-agent tokens up to itself: state token counted in Ar already; rtg/action add kq
    for qblk in range(nqb):
        qb=qblk*QB
        tqs=[]
        for w in range(nw):
            qs=[min(qb+w*WQ+l, Lq-1) for l in range(WQ)]
            live = qb+w*WQ < Lq
            tq=[((p-rep_pos0)//3 if (rep_keys and p>=rep_pos0) else p//A3) for p in qs]
            tqs.append((min(tq),max(tq),live))
        bt=max(t[1] for t in tqs)
        k_end=min(Lk,(bt+1)*A3); rep_need=min(rep_keys,(bt+1)*3)
        n_reg=(k_end+KT-1)//KT; n_rep=(rep_need+KT-1)//KT; n_it=n_reg+n_rep
        tot['tiles_wg']+=n_it
        t_last=(Lk-1)//A3
        for it in range(n_it):
            costs=[]
            for (tmin,tmax,live) in tqs:
                c=0.0
                if not live: costs.append(0); continue
                for sub in range(KT//32):
                    if it<n_reg:
                        ks0=it*KT+sub*32
                        t_lo=ks0//A3; t_hi=min((ks0+31)//A3,t_last)
                        if ks0>=k_end or t_lo>tmax: continue
                        need=not (t_hi<tmin and ks0+31<Lk)
                    else:
                        j0=(it-n_reg)*KT+sub*32
                        if j0>=rep_need or j0>3*tmax+2: continue
                        need=not (j0+31<=3*tmin and j0+32<=rep_keys)
                    if need: tot['msk']+=1; c+=1.6
                    else: tot['unm']+=1; c+=1.0
                costs.append(c)
            tot['crit']+=max(costs)
            tot['wave_tiles']+=nw
    ex=tot['unm']+tot['msk']
    return dict(Actx=Actx,Lq=Lq,wgs=nqb,exec=ex,useful=useful/1024,eff=useful/1024/ex,msk_frac=tot['msk']/ex,
                tiles_wg=tot['tiles_wg'],crit=tot['crit'], wsum=tot['unm']+1.6*tot['msk'])
if __name__=='__main__':
    for QB,WQ in ((128,32),(64,32),(64,16)):
        print('QB',QB,'WQ',WQ)
        for A in (4,6,8,10,12,14,16,20):
            r=sched(A,QB,WQ)
            print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items()})
        r=sched(24,QB,WQ,plain=True); print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items()})
