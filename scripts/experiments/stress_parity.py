"""Randomised parity stress against the numpy oracle over tile geometries with several waves per workgroup (developer tool; GPU)."""
import sys; import os; ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'qhbm-library_amd'))
import numpy as np
from oracle import qhbm_oracle as O
from qhbmlib_amd import _engine as E
from tests.test_engine_gpu import random_circuit, _engine, check_values, check_jacobian
bad=0
first=int(sys.argv[2]) if len(sys.argv) > 2 else 0
for seed in range(first, first+(int(sys.argv[1]) if len(sys.argv) > 1 else 24)):
  rng=np.random.default_rng(1000+seed)
  n=13+seed%2
  if seed%3==0:
    gates,names=O.hea_gates(n, 3+seed%3, "s"); P=len(names)
  else:
    P=8; gates=random_circuit(rng,n,70,P)
  params=rng.uniform(-1,1,P)
  ops=[O.random_pauli_op(n,10,seed,p_identity=0.7), O.xxz_chain_op(n)]
  if seed%2==1: ops=[ops[0]+ops[1]]   # one observable: the values come out of lambda = O psi in the VJP calls
  bits=rng.integers(0,2,size=(2,n)).astype(np.int8)
  for tile,adj in ((12,12),(11,11),(13,12),(10,10),(12,13)):
    eng=_engine(n,gates,P,ops,tile_qubits=min(tile,n),adjoint_tile_qubits=min(adj,n-1), adjoint_exchange=seed%4!=3)
    try:
      check_values(eng,n,gates,params,bits,ops,rel=3e-5)
      check_jacobian(eng,n,gates,params,bits[:1],ops,rel=3e-4)
    except AssertionError as e:
      bad+=1; print("FAIL seed",seed,"tile",tile,adj,str(e)[:120])
print("done, failures:",bad)
