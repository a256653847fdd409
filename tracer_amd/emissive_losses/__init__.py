"""
Thermal-emission losses of cavity receivers: ray-traced view factors and the radiosity solve
(the reference's `emissive_losses` package).
"""
