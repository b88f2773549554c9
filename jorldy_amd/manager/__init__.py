from .batched_actors import BatchedValueActors, VecNStepApeX
from .collector import NativeCollector, VecCollector

__all__ = ["VecCollector", "NativeCollector", "BatchedValueActors", "VecNStepApeX"]
