"""Counterparts of the reference's ``dynamont.segmentation`` package (segment.py, train.py,
utils.py): same CLI flags, CSV bytes, ``.errors`` format and model-file format, driving the
MI355X core in batches."""
