from .coders import AABBCoder, MidpointOffsetCoder, BaseBBoxCoder  # noqa: F401
