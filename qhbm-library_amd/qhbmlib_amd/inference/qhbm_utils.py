"""Utilities for metrics on QHBMs (reference: qhbmlib/inference/qhbm_utils.py)."""
import torch

from qhbmlib_amd.inference import ebm_utils
from qhbmlib_amd.inference import qnn_utils
from qhbmlib_amd.models import hamiltonian


def density_matrix(model: hamiltonian.Hamiltonian):
  """Thermal state of a modular Hamiltonian, rho = U_phi P_theta U_phi^dagger
  (qhbm_utils.py:24-59)."""
  unitary_matrix = qnn_utils.unitary(model.circuit)
  probs = ebm_utils.probabilities(model.energy).to(unitary_matrix.device, torch.complex64)
  return torch.einsum("k,ik,kj->ij", probs, unitary_matrix, unitary_matrix.conj().transpose(0, 1))


def fidelity(model: hamiltonian.Hamiltonian, sigma: torch.Tensor):
  """Fidelity (tr sqrt(sqrt(rho) sigma sqrt(rho)))^2 between a QHBM and a density matrix, through
  the Hermitian omega = sqrt(K) U^dagger sigma U sqrt(K) (qhbm_utils.py:62-116)."""
  u_phi = qnn_utils.unitary(model.circuit).to(torch.complex128)
  sigma = torch.as_tensor(sigma).to(u_phi.device, torch.complex128)
  k_theta = ebm_utils.probabilities(model.energy).to(u_phi.device, torch.complex128)
  sqrt_k_theta = torch.sqrt(k_theta)
  # omega and its spectrum in double precision: sqrt() turns eigenvalue noise eps into sqrt(eps),
  # which in complex64 is the reference's whole rtol of 1e-4
  omega = torch.einsum("a,ab,bc,cd,d->ad", sqrt_k_theta, u_phi.conj().transpose(0, 1), sigma, u_phi,
                       sqrt_k_theta)
  d_omega = torch.linalg.eigvalsh(omega)
  return (torch.sum(torch.sqrt(d_omega.clamp_min(0.0)))**2).to(torch.float32)
