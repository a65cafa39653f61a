VARIANTS = {"generic": {"conv3": 0, "hconv": 0}, "hconv": {"conv3": 0, "hconv": 1}}
