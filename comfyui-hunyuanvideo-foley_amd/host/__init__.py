"""Host-side (Python) half of the MI355X Foley sampling path."""
