"""Train / test loop helpers mirroring the reference's lib/helpers/* API (same function and class
names, argument meaning and checkpoint dictionary)."""
