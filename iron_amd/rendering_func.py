"""placeholder - filled in with get_materials below"""
