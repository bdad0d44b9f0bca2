"""Randomised parity stress of the values MEASURED IN THE PASSES under default plan options (wide last pass, measurement-only
passes, the Walsh-Hadamard path for >= 32 diagonal terms) against the numpy oracle, 15..20 qubits (developer tool; GPU).
  python scripts/experiments/stress_measure.py [seeds] [first seed]      (STRESS_N_BASE / STRESS_N_SPAN: qubit counts, default 15 / 6)
Round 5 added it after the wide-last-pass / WHT class bug: the fixed-tile stress of stress_parity.py never plans a wide pass."""
import sys; import os; ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'qhbm-library_amd'))
import numpy as np
from oracle import qhbm_oracle as O
from qhbmlib_amd import _engine as E
bad=0
first=int(sys.argv[2]) if len(sys.argv) > 2 else 0
for seed in range(first, first+(int(sys.argv[1]) if len(sys.argv) > 1 else 18)):
  rng=np.random.default_rng(5000+seed)
  n=int(os.environ.get('STRESS_N_BASE','15'))+seed%int(os.environ.get('STRESS_N_SPAN','6'))
  gates,names=O.hea_gates(n, 2+seed%3, "sm"); P=len(names)
  params=rng.uniform(-1,1,P)
  k=int(rng.integers(20,90))                      # diagonal terms: below and above the WHT threshold (32)
  def zstring():
    w=int(rng.integers(1,4)); qs=rng.choice(n,size=w,replace=False)
    return (float(rng.normal()),0,int(sum(1<<int(q) for q in qs)))
  diag=[zstring() for _ in range(k)]
  if seed%2==0:  # short-range strings (what the LAST pass of the sweep can measure itself: the wide-pass case)
    diag=[(float(rng.normal()),0,(1<<q)|(1<<((q+1)%n))) for q in range(n)]+[(float(rng.normal()),0,1<<q) for q in range(n)]
    diag+= [(float(rng.normal()),0,(1<<q)|(1<<((q+2)%n))) for q in range(int(rng.integers(0,n)))]
  flips=[(float(rng.normal()),1<<int(q),0) for q in rng.choice(n,size=int(rng.integers(0,6)),replace=False)]
  flips+=[(float(rng.normal()),(1<<int(a))|(1<<int(b)),1<<int(a)) for a,b in [rng.choice(n,size=2,replace=False) for _ in range(int(rng.integers(0,3)))]]
  layout=seed%3
  if layout==0: ops=[[t] for t in diag]+([flips] if flips else [])          # shards
  elif layout==1:
    m=int(rng.integers(2,6)); ops=[diag[i::m]+flips[i::m] for i in range(m)]  # a few observables with many terms each
  else: ops=[diag+flips, [zstring() for _ in range(int(rng.integers(1,40)))]]
  ops=[op for op in ops if op]
  bits=rng.integers(0,2,size=(2,n)).astype(np.int8)
  want=O.expectation(n,gates,params,bits,ops)
  for opts in ({}, {"observable_kernel":0,"multi_observable_values":0}, {"wide_last_pass":0}):
    eng=E.Engine(0)
    for kk,v in opts.items(): eng.set_option(kk,v)
    eng.set_circuit(n,gates,P); eng.set_observables(ops)
    got=eng.expectation(bits,params).cpu().numpy()
    norm=np.array([sum(abs(c) for c,_,_ in op) for op in ops])
    err=np.abs(got-want)/np.maximum(norm,1.0)[None,:]
    if err.max()>5e-5:
      bad+=1; print("FAIL seed",seed,"n",n,"layout",layout,"k",k,"opts",opts,"max rel err",float(err.max()))
print("done, failures:",bad)
