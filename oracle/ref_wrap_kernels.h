// kernel wrappers (filled in below)
