"""Model classes mirroring the reference's `models/` API for the hot path."""


def create_model(opt, rank):
    """models/__init__.py:79-93 of the reference, restricted to the model types built here."""
    if opt.model_type == "palette":
        from .palette_model import PaletteModel

        return PaletteModel(opt, rank)
    if opt.model_type == "cm":
        from .cm_model import CMModel

        return CMModel(opt, rank)
    if opt.model_type == "cut":
        from .cut_model import CUTModel

        return CUTModel(opt, rank)
    raise NotImplementedError(f"model_type {opt.model_type!r} is not implemented in joligen_amd yet")
