"""VQT loss (reference: qhbmlib/inference/vqt_loss.py)."""
import torch

from qhbmlib_amd.inference import qhbm  # noqa: F401


def vqt(input_qhbm: "qhbm.QHBM", target_hamiltonian, beta):
  """beta <H> - S(rho) as a differentiable sample average (vqt_loss.py:25-55).
  `target_hamiltonian` is `[pauli_sum]` (one operator) or a Hamiltonian."""
  beta = torch.as_tensor(beta, dtype=torch.float32)
  if not beta.is_cuda and not beta.requires_grad:
    beta = float(beta)   # a host scalar multiplies as a kernel argument: no host-to-device copy per step (none under hipGraph capture)

  def f_vqt(bitstrings):
    h_expectations = torch.squeeze(
        input_qhbm.q_inference.expectation(bitstrings, target_hamiltonian), 1)
    beta_h_expectations = (beta if isinstance(beta, float) else beta.to(h_expectations.device)) * h_expectations
    energies = input_qhbm.e_inference.energy(bitstrings).detach()
    return beta_h_expectations - energies.to(h_expectations.device)

  input_qhbm.agree_seeds()   # (a no-op unless the quantum inference is sharded over a process group)
  average_expectation = input_qhbm.e_inference.expectation(f_vqt)
  current_partition = input_qhbm.e_inference.log_partition().detach()
  return average_expectation - current_partition.to(average_expectation.device)
