"""Topologies the B-cos layers are wired into.  Nothing is imported eagerly."""
