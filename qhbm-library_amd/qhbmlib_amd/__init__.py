"""MI355X-native expectation engine behind the qhbmlib QuantumInference API."""
from qhbmlib_amd import _engine  # noqa: F401
from qhbmlib_amd import data  # noqa: F401
from qhbmlib_amd import inference  # noqa: F401
from qhbmlib_amd import ir  # noqa: F401
from qhbmlib_amd import models  # noqa: F401
from qhbmlib_amd import utils  # noqa: F401
from qhbmlib_amd._engine import Engine, EngineError  # noqa: F401

__version__ = "0.1.0"
