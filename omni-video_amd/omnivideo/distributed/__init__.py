"""Sequence / context parallelism of the sampling path (reference: omnivideo/distributed/).
FSDP weight sharding (reference fsdp.py) is not provided: both experts fit one MI355X."""
