"""Randomised parity stress of values AND gradients under the engine's DEFAULT plan choices (tile sizes, wide last pass,
relabeling adjoint, pairs, observable kernel by cost model) against the numpy oracle, 15..18 qubits (developer tool; GPU).
  python scripts/experiments/stress_default_plans.py [seeds] [first seed]
Round 5: stress_parity.py pins the tile sizes at 13 / 14 qubits and so never sees what the planner picks on its own."""
import sys; import os; ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'qhbm-library_amd'))
import numpy as np
from oracle import qhbm_oracle as O
from qhbmlib_amd import _engine as E
from tests.test_engine_gpu import random_circuit
bad=0
first=int(sys.argv[2]) if len(sys.argv) > 2 else 0
for seed in range(first, first+(int(sys.argv[1]) if len(sys.argv) > 1 else 12)):
  rng=np.random.default_rng(7000+seed)
  n=15+seed%4
  if seed%2==0:
    gates,names=O.hea_gates(n, 2+seed%3, "sd"); P=len(names)
  else:
    P=10; gates=random_circuit(rng,n,60+10*(seed%3),P)
  params=rng.uniform(-1,1,P)
  pool=[O.random_pauli_op(n,12,seed,p_identity=0.75), O.xxz_chain_op(n), O.tfim_ring_op(n),
        [(float(rng.normal()),0,(1<<q)|(1<<((q+1)%n))) for q in range(n)]+[(float(rng.normal()),0,1<<q) for q in range(n)]]
  ops=[pool[i] for i in rng.choice(4,size=1+seed%3,replace=False)]
  bits=rng.integers(0,2,size=(2,n)).astype(np.int8)
  up=rng.normal(size=(2,len(ops)))
  want_vals,want_jac=O.expectation_jacobian(n,gates,params,bits,ops)
  want_grad=np.einsum("bt,btp->p",up,want_jac)
  norm=np.array([sum(abs(c) for c,_,_ in op) for op in ops])
  for opts in ({}, {"adjoint_relabel":0}, {"forward_pairs":0,"wide_last_pass":0}, {"adjoint_tile_qubits":13}):
    eng=E.Engine(0)
    for kk,v in opts.items(): eng.set_option(kk,v)
    eng.set_circuit(n,gates,P); eng.set_observables(ops)
    mask=None
    if seed%5==3:
      mask=rng.random(P)<0.6; eng.set_gradient_mask(mask)
    try:
      got=eng.expectation(bits,params).cpu().numpy()
      vals,grad=eng.expectation_vjp(bits,params,up)
      ev=max(float((np.abs(got-want_vals)/np.maximum(norm,1.0)[None,:]).max()), float((np.abs(vals.cpu().numpy()-want_vals)/np.maximum(norm,1.0)[None,:]).max()))
      g=grad.cpu().numpy(); wg=want_grad if mask is None else np.where(mask,want_grad,0.0)
      eg=float(np.abs(g-wg).max()/max(1.0,np.abs(wg).max()))
      if ev>5e-5 or eg>3e-4:
        bad+=1; print("FAIL seed",seed,"n",n,"opts",opts,"mask",mask is not None,"value err",ev,"grad err",eg)
    except Exception as e:  # pylint: disable=broad-except
      bad+=1; print("ERROR seed",seed,"n",n,"opts",opts,repr(e)[:200])
print("done, failures:",bad)
