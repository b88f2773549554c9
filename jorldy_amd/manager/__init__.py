from .batched_actors import BatchedValueActors, DeviceActorFeed, VecNStepApeX
from .collector import NativeCollector, VecCollector

__all__ = ["VecCollector", "NativeCollector", "BatchedValueActors", "VecNStepApeX", "DeviceActorFeed"]
