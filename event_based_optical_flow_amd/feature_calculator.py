"""The reference ships only a mock feature calculator (src/feature_calculator.py); warp_event
therefore returns (warped, feature_dict) 2-tuples.  This keeps that return shape."""


def skip_feature() -> dict:
    return {"none": {"per_event": True, "value": None}}
