"""Models: circuits, energies, Hamiltonians (reference: qhbmlib/models/__init__.py:29-41)."""
from qhbmlib_amd.models.circuit import DirectQuantumCircuit, QAIA, QuantumCircuit
from qhbmlib_amd.models.energy import BernoulliEnergy, BitstringEnergy, KOBE, PauliMixin
from qhbmlib_amd.models.energy_utils import Parity, SpinsFromBitstrings, VariableDot
from qhbmlib_amd.models.hamiltonian import Hamiltonian
from qhbmlib_amd.models import circuit_utils  # noqa: F401

__all__ = ["BernoulliEnergy", "BitstringEnergy", "DirectQuantumCircuit", "Hamiltonian", "KOBE",
           "Parity", "PauliMixin", "QAIA", "QuantumCircuit", "SpinsFromBitstrings", "VariableDot"]
