"""Conveniences outside the SURVEY section 8 scope (see each module's header)."""
