"""Quantum data (reference: qhbmlib/data/quantum_data.py:25-41, qhbm_data.py:26-38)."""
import abc


class QuantumData(abc.ABC):
  """Interface for quantum datasets."""

  @abc.abstractmethod
  def expectation(self, observable):
    """Average of `observable` against this data source."""
    raise NotImplementedError()


class QHBMData(QuantumData):
  """QuantumData defined by a QHBM (qhbm_data.py:26-38)."""

  def __init__(self, input_qhbm):
    self.qhbm = input_qhbm

  def expectation(self, observable):
    return self.qhbm.expectation(observable).squeeze(0)
