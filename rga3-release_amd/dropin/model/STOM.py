from rga3.model.STOM import STOM  # noqa: F401
