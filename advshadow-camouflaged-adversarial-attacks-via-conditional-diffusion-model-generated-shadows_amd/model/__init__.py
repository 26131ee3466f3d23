"""Mirror of the reference's ``model`` package for the hot path (networks + samplers)."""
