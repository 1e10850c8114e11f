"""Local counterparts of tidy3d plugins that sit on the solver's path."""
